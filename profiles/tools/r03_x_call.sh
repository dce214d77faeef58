#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 500 python profiles/tools/xcd_sweep.py 12 14 18 22 26 30 32 34 36 38 40 44 48 > $OUT/xcd_sweep2.txt 2>&1; rc=$?; cat $OUT/xcd_sweep2.txt | tail -30; echo "rc=$rc"
