#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 600 python profiles/tools/xcd_fuzz.py 60 1 > $OUT/xcd_fuzz.txt 2>&1; rc=$?; tail -70 $OUT/xcd_fuzz.txt; echo "rc=$rc"
