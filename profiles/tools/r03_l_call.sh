#!/bin/bash
OUT=gpurun_out/r03_l; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_multiproc.py -q -x -k "failing_rank" > $OUT/pytest_inject.log 2>&1; rc=$?; echo "pytest inject rc=$rc"; tail -5 $OUT/pytest_inject.log | cut -c1-300
cat gpurun_out/multiproc_last_failure.log 2>/dev/null | cut -c1-400 | tail -60
echo finished
