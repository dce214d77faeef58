#!/bin/bash
# round 3, call k: multi-process tests (injection over three routes), rocprofv3 evidence of the final kernels, default bench line
OUT=gpurun_out/r03_k; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py -q -x > $OUT/pytest_multiproc.log 2>&1; rc=$?; echo "pytest multiproc rc=$rc"; tail -5 $OUT/pytest_multiproc.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
NEUTFEM_COMMIT=$(cat profiles/tools/commit.txt 2>/dev/null) timeout -k 10 900 bash profiles/collect.sh r03_k > $OUT/collect.log 2>&1; rc=$?; echo "collect rc=$rc"; tail -6 $OUT/collect.log | cut -c1-200
[ $rc -eq 124 ] && exit 1
cp gpurun_out/prof_r03_k/r03_k_* $OUT/ 2>/dev/null; cp gpurun_out/prof_r03_k/bench_stats.json $OUT/bench_under_rocprof.json 2>/dev/null; rm -rf gpurun_out/prof_r03_k/stats gpurun_out/prof_r03_k/fetch gpurun_out/prof_r03_k/write
mkdir -p profiles; cp $OUT/r03_k_pmc_traffic.json profiles/r03_k_pmc_traffic_256cube.json     # so that the bench line below finds the matching profile
t0=$(date +%s); timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_k/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['avg_ms'], r['traffic'], r['traffic_source'].get('kernel') if r['traffic_source'] else None, [(p['name'],p['avg_ms'],p['achieved']) for p in r['passes']])
print(json.dumps(d.get('c5_single_gpu'))[:700])
PY
echo finished
