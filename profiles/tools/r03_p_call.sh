#!/bin/bash
# round 3, call p: rocprofv3 evidence of the final kernels, default bench line, full GPU suite
OUT=gpurun_out/r03_p; mkdir -p $OUT
NEUTFEM_COMMIT=$(cat profiles/tools/commit.txt 2>/dev/null) timeout -k 10 900 bash profiles/collect.sh r03_p > $OUT/collect.log 2>&1; rc=$?; echo "collect rc=$rc"; tail -6 $OUT/collect.log | cut -c1-200
[ $rc -eq 124 ] && exit 1
cp gpurun_out/prof_r03_p/r03_p_* $OUT/ 2>/dev/null; cp gpurun_out/prof_r03_p/bench_stats.json $OUT/bench_under_rocprof.json 2>/dev/null; rm -rf gpurun_out/prof_r03_p/stats gpurun_out/prof_r03_p/fetch gpurun_out/prof_r03_p/write
cp $OUT/r03_p_pmc_traffic.json profiles/r03_p_pmc_traffic_256cube.json
t0=$(date +%s); timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_p/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['avg_ms'], r['traffic'], [(p['name'],p['avg_ms'],p['achieved']) for p in r['passes']])
print(json.dumps(d.get('c5_single_gpu'))[:900])
for c in d.get('other_configs', []): print(c['config'][:40], c['solve_ms'], c['flux_rel_l2_vs_oracle'], c['pcm_vs_oracle'])
PY
( while true; do sleep 60; echo "[alive $(date +%H:%M:%S)] $(tail -c 200 $OUT/pytest.log 2>/dev/null | tr '\n' ' ' | tail -c 100)"; done ) &
HB=$!
timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/pytest.log | cut -c1-250
kill $HB 2>/dev/null
echo finished
