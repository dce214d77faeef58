#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT
(while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)] $(tail -c 100 $OUT/pytest_dur.txt 2>/dev/null | tr '\n' ' ')"; done) &
HB=$!
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=60 > $OUT/pytest_dur.txt 2>&1; rc=$?
kill $HB
tail -75 $OUT/pytest_dur.txt; echo "rc=$rc"
