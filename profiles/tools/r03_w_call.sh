#!/bin/bash
# barrier cost on one XCD (profiles/tools/xcd_barrier.hip): participants per XCD x threads per workgroup
OUT=gpurun_out/r03_w; mkdir -p $OUT; rm -f $OUT/xcd_barrier2.txt
for cfg in "256 1024" "256 256" "512 256" "768 256" "384 512"; do
  timeout -k 10 120 ./profiles/tools/xcd_barrier $cfg 2000 >> $OUT/xcd_barrier2.txt 2>&1; rc=$?; echo "cfg $cfg rc=$rc" >> $OUT/xcd_barrier2.txt
  [ $rc -ne 0 ] && break
done
grep -v "mode 2" $OUT/xcd_barrier2.txt
