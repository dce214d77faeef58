#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_xcd.py -x -q -m gpu > $OUT/pytest_xcd.txt 2>&1; rc=$?; tail -25 $OUT/pytest_xcd.txt; echo "xcd tests rc=$rc"
