#!/bin/bash
# round 3, call j: full GPU suite with durations, default bench line
OUT=gpurun_out/r03_j; mkdir -p $OUT
( while true; do sleep 60; echo "[alive $(date +%H:%M:%S)] $(tail -c 200 $OUT/pytest.log 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done ) &
HB=$!
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=30 > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -45 $OUT/pytest.log | cut -c1-200
kill $HB 2>/dev/null
[ $rc -eq 124 ] && exit 1
t0=$(date +%s); timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_j/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['avg_ms'], r['traffic'], [(p['name'],p['avg_ms'],p['achieved']) for p in r['passes']])
print(json.dumps(d.get('c5_single_gpu'))[:900])
for c in d.get('other_configs', []): print(c['config'][:40], c['solve_ms'], c['flux_rel_l2_vs_oracle'], c['pcm_vs_oracle'])
PY
echo finished
