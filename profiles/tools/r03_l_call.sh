#!/bin/bash
OUT=gpurun_out/r03_l; mkdir -p $OUT; rm -f gpurun_out/multiproc_last_failure.log
timeout -k 10 500 python -m pytest tests/test_gpu_multiproc.py -q -x > $OUT/pytest_multiproc.log 2>&1; rc=$?; echo "pytest multiproc rc=$rc"; tail -4 $OUT/pytest_multiproc.log | cut -c1-200
timeout -k 10 400 python -m pytest tests/test_gpu_longlines.py -q -x > $OUT/pytest_long.log 2>&1; rc=$?; echo "pytest long rc=$rc"; tail -4 $OUT/pytest_long.log | cut -c1-200
bash profiles/tools/size_sweep.sh $OUT/sweep > $OUT/sweep.txt 2>&1; echo "sweep rc=$?"; cat $OUT/sweep.txt
echo finished
