#!/bin/bash
# round 3, call g: multi-process tests first (vector reduce, injected failure), then the full GPU suite
OUT=gpurun_out/r03_g; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py -q -x > $OUT/pytest_multiproc.log 2>&1; rc=$?; echo "pytest multiproc rc=$rc"; tail -30 $OUT/pytest_multiproc.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
( while true; do sleep 60; echo "[alive $(date +%H:%M:%S)] $(tail -c 200 $OUT/pytest.log 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
HB=$!
timeout -k 10 1000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multiproc.py > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/pytest.log | cut -c1-400
kill $HB 2>/dev/null
echo finished
