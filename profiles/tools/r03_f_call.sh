#!/bin/bash
# round 3, call f: 256^3 golden test, A/B inside CG at 512^3 after the partial-sum fix, path tests, default bench line
OUT=gpurun_out/r03_f; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "256cube_golden or 128cube_golden or solve_keff_golden" > $OUT/pytest_golden.log 2>&1; rc=$?; echo "pytest golden rc=$rc"; grep -E "^256|^128|same_path|passed|failed" $OUT/pytest_golden.log | cut -c1-250
[ $rc -eq 124 ] && exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_paths.py tests/test_gpu_longlines.py -q -x > $OUT/pytest_paths.log 2>&1; rc=$?; echo "pytest paths rc=$rc"; tail -3 $OUT/pytest_paths.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
timeout -k 10 500 python profiles/tools/ab_cg.py checker 512 2 default split_dot=0 x_two_phase=1 > $OUT/ab_cg_512.txt 2>&1; rc=$?; echo "ab_cg 512 rc=$rc"; cat $OUT/ab_cg_512.txt | cut -c1-220
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python profiles/tools/ab_cg.py iaea3d 256 2 default split_dot=2 > $OUT/ab_cg_256.txt 2>&1; rc=$?; echo "ab_cg 256 rc=$rc"; cat $OUT/ab_cg_256.txt | cut -c1-220
[ $rc -eq 124 ] && exit 1
t0=$(date +%s); timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_f/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['avg_ms'], [(p['name'],p['avg_ms'],p['achieved']) for p in r['passes']])
print(json.dumps(d.get('c5_single_gpu'))[:1500])
for c in d.get('other_configs', []): print(c['config'][:40], c['solve_ms'], c['flux_rel_l2_vs_oracle'], c['pcm_vs_oracle'])
PY
echo finished
