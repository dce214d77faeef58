#!/bin/bash
# round 3, call e: A/B inside CG (x two-phase loads, split dot, chunked lines) at 256^3 and 512^3 x 2 groups, then the full GPU suite
OUT=gpurun_out/r03_e; mkdir -p $OUT
timeout -k 10 400 python profiles/tools/ab_cg.py iaea3d 256 2 default x_two_phase=0 x_two_phase=1 split_dot=0 nt_loads=0 > $OUT/ab_cg_256.txt 2>&1; rc=$?; echo "ab_cg 256 rc=$rc"; cat $OUT/ab_cg_256.txt | cut -c1-220
[ $rc -eq 124 ] && exit 1
timeout -k 10 500 python profiles/tools/ab_cg.py checker 512 2 default x_two_phase=0 x_two_phase=1 split_dot=0,s_long=0 s_long=0 > $OUT/ab_cg_512.txt 2>&1; rc=$?; echo "ab_cg 512 rc=$rc"; cat $OUT/ab_cg_512.txt | cut -c1-220
[ $rc -eq 124 ] && exit 1
( while true; do sleep 60; echo "[alive $(date +%H:%M:%S)] $(tail -c 200 $OUT/pytest.log 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
HB=$!
timeout -k 10 1000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_parity.py::test_iaea3d_256cube_golden > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $OUT/pytest.log | cut -c1-400
kill $HB 2>/dev/null
echo finished
