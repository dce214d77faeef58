#!/bin/bash
OUT=gpurun_out/r03_q; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_longlines.py tests/test_gpu_paths.py -q -x --durations=5 > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/pytest.log | cut -c1-200
echo finished
