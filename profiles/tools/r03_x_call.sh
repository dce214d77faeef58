#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 300 python profiles/tools/xcd_stress.py 60 > $OUT/xcd_stress.txt 2>&1; rc=$?; cat $OUT/xcd_stress.txt | tail -30; echo "rc=$rc"
