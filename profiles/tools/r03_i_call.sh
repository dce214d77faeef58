#!/bin/bash
# round 3, call i: device quadrature assembly (tests + FMA vs MFMA timing), A/B of chunked y / y+z passes inside CG at 256^3, 192^3, 128^3
OUT=gpurun_out/r03_i; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_assembly.py -q -x > $OUT/pytest_assembly.log 2>&1; rc=$?; echo "pytest assembly rc=$rc"; tail -12 $OUT/pytest_assembly.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python profiles/tools/assembly_bench.py 16384 > $OUT/assembly_bench.txt 2>&1; echo "assembly bench rc=$?"; cat $OUT/assembly_bench.txt | cut -c1-220
for n in 256 192 128; do
  timeout -k 10 400 python profiles/tools/ab_cg.py iaea3d $n 2 default s_long=1,s_long_dirs=1 s_long=1,s_long_dirs=3 s_long=1,s_long_dirs=2 > $OUT/ab_cg_$n.txt 2>&1; rc=$?; echo "ab_cg $n rc=$rc"; cat $OUT/ab_cg_$n.txt | cut -c1-220
  [ $rc -eq 124 ] && exit 1
done
echo finished
