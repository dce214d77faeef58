#!/bin/bash
# round 3, call c: long-line parity (non-spilling chunked kernel), A/B at 512^3, default bench line (256^3 headline + small configs + C5 with self-check)
OUT=gpurun_out/r03_c; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_longlines.py -q -x > $OUT/pytest_long.log 2>&1; rc=$?; echo "pytest long rc=$rc"; tail -3 $OUT/pytest_long.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
timeout -k 10 500 python profiles/tools/ab_long.py checker 512 2 10 > $OUT/ab_512.txt 2>&1; rc=$?; echo "ab 512 rc=$rc"; head -9 $OUT/ab_512.txt | cut -c1-200
[ $rc -eq 124 ] && exit 1
t0=$(date +%s); timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_c/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['avg_ms'], [(p['name'],p['avg_ms'],p['achieved']) for p in r['passes']])
print(json.dumps(d.get('c5_single_gpu'))[:1500])
for c in d.get('other_configs', []): print(c['config'][:40], c['solve_ms'], c['flux_rel_l2_vs_oracle'], c['pcm_vs_oracle'])
PY
echo finished
