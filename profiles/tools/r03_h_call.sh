#!/bin/bash
# round 3, call h: multi-process tests again (prepare-time collective), rocprofv3 evidence of the final kernels (collect.sh), PMC traffic of the
# chunked kernels at 512^3, loopback slabs for the scaling model, A/B of the line kernels at 256^3 with the final chunked kernel
OUT=gpurun_out/r03_h; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_multiproc.py -q -x > $OUT/pytest_multiproc.log 2>&1; rc=$?; echo "pytest multiproc rc=$rc"; tail -3 $OUT/pytest_multiproc.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
NEUTFEM_COMMIT=$(cat profiles/tools/commit.txt 2>/dev/null) timeout -k 10 900 bash profiles/collect.sh r03_h > $OUT/collect.log 2>&1; rc=$?; echo "collect rc=$rc"; tail -12 $OUT/collect.log | cut -c1-200
[ $rc -eq 124 ] && exit 1
cp gpurun_out/prof_r03_h/r03_h_* $OUT/ 2>/dev/null; cp gpurun_out/prof_r03_h/bench_stats.json $OUT/bench_under_rocprof.json 2>/dev/null; rm -rf gpurun_out/prof_r03_h/stats gpurun_out/prof_r03_h/fetch gpurun_out/prof_r03_h/write
bash profiles/tools/pmc_groups.sh r03_h_512_chunked checker 512 512 512 2 4 || exit 1
python3 profiles/tools/pmc_summarize.py gpurun_out/pmc_r03_h_512_chunked $OUT/pmc_512_chunked.json "512^3 x 2 groups, final chunked long-line kernels (32-bit offsets, z.w dot)" > $OUT/pmc_512_chunked_summary.txt 2>&1
rm -rf gpurun_out/pmc_r03_h_*/g*/ 2>/dev/null
timeout -k 10 300 python profiles/tools/ab_long.py iaea3d 256 2 20 > $OUT/ab_256.txt 2>&1; echo "ab 256 rc=$?"; head -9 $OUT/ab_256.txt | cut -c1-200
for lb in 8 2; do
  timeout -k 10 300 python bench.py --loopback-slabs $lb --steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/bench_256_loopback$lb.json 2> $OUT/lb$lb.err; echo "loopback $lb rc=$?"
done
timeout -k 10 400 python bench.py --case checker --n 512 --groups 2 --loopback-slabs 8 --steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/bench_512_loopback8.json 2> $OUT/lb512.err; echo "loopback 512 rc=$?"
timeout -k 10 400 python bench.py --case checker --n 512 --groups 2 --steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/bench_512_g2.json 2> $OUT/b512.err; echo "512 g2 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_h/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['cg_iters_per_outer'], [(p['name'],round(p['avg_ms']*1e3,1)) for p in r['passes']])
    except Exception as e: print(f, 'ERR', e)
PY
echo finished
