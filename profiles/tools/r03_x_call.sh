#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 300 python profiles/tools/xcd_stress.py 150 1 iaea2d 1 > $OUT/xcd_stress_rt1.txt 2>&1; rc=$?; tail -4 $OUT/xcd_stress_rt1.txt; echo "rc=$rc"
