#!/bin/bash
# robustness: the whole GPU suite with the chunked long-line kernel forced onto every eligible y / z line (NEUTFEM_OPTS)
OUT=gpurun_out/r03_r; mkdir -p $OUT
( while true; do sleep 60; echo "[alive $(date +%H:%M:%S)] $(tail -c 200 $OUT/pytest.log 2>/dev/null | tr '\n' ' ' | tail -c 100)"; done ) &
HB=$!
NEUTFEM_OPTS="s_long=1" timeout -k 10 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multiproc.py > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $OUT/pytest.log | cut -c1-220
kill $HB 2>/dev/null
echo finished
