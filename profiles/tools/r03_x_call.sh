#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_paths.py -x -q -m gpu > $OUT/pytest_paths.txt 2>&1; rc=$?; tail -5 $OUT/pytest_paths.txt; echo "paths rc=$rc"
