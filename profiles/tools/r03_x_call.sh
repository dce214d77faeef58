#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
timeout -k 10 400 python profiles/tools/xcd_sweep.py > $OUT/xcd_sweep.txt 2>&1; rc=$?; cat $OUT/xcd_sweep.txt | tail -30; echo "rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 240 python profiles/tools/xcd_dev.py 7 > $OUT/xcd_dev.txt 2>&1; rc=$?; cat $OUT/xcd_dev.txt | tail -30; echo "rc=$rc"
