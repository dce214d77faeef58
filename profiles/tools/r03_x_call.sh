#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 300 python profiles/tools/xcd_stress.py 150 1 > $OUT/xcd_stress_whole.txt 2>&1; rc=$?; tail -4 $OUT/xcd_stress_whole.txt; echo "rc=$rc"; [ $rc -eq 124 ] && exit 1
timeout -k 10 300 python profiles/tools/xcd_stress.py 100 0 > $OUT/xcd_stress_cg.txt 2>&1; rc=$?; tail -4 $OUT/xcd_stress_cg.txt; echo "rc=$rc"
