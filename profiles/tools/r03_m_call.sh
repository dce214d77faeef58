#!/bin/bash
OUT=gpurun_out/r03_m; mkdir -p $OUT
( while true; do sleep 60; echo "[alive $(date +%H:%M:%S)] $(tail -c 200 $OUT/pytest.log 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
HB=$!
timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $OUT/pytest.log | cut -c1-250
kill $HB 2>/dev/null
echo finished
