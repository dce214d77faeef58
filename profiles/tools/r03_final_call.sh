#!/bin/bash
# closing check: smoke(), the whole GPU suite in one process
OUT=gpurun_out/r03_final; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; tail -2 $OUT/smoke.txt; echo "smoke rc=$rc"; [ $rc -ne 0 ] && exit 1
(while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)] $(tail -c 120 $OUT/pytest_all.txt 2>/dev/null | tr '\n' ' ')"; done) &
HB=$!
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1; rc=$?
kill $HB
tail -8 $OUT/pytest_all.txt; echo "all rc=$rc"
