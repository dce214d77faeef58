#!/bin/bash
# k_cg_xcd on by default: the path tests first, then the whole GPU suite (heartbeat lines for the silence guard)
OUT=gpurun_out/r03_y; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_paths.py -x -q -m gpu > $OUT/pytest_paths.txt 2>&1; rc=$?; tail -15 $OUT/pytest_paths.txt; echo "paths rc=$rc"
[ $rc -ne 0 ] && exit 1
(while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)] $(tail -c 200 $OUT/pytest_all.txt 2>/dev/null | tr '\n' ' ')"; done) &
HB=$!
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_paths.py > $OUT/pytest_all.txt 2>&1; rc=$?
kill $HB
tail -15 $OUT/pytest_all.txt; echo "all rc=$rc"
