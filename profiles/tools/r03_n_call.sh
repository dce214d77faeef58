#!/bin/bash
# round 3, call n: tile-shape A/B on thin slabs (256^3 as 8 slabs on one GPU), new direct / long-line tests
OUT=gpurun_out/r03_n; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_direct.py tests/test_gpu_longlines.py -q -x > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log | cut -c1-200
for opts in "" "s_tx=16" "s_tx=32" "s_long=0" "s_long=0,s_tx=16" "s_long_dirs=0"; do
  NEUTFEM_OPTS="$opts" timeout -k 10 300 python bench.py --loopback-slabs 8 --steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/lb8.json 2> $OUT/lb8.err; rc=$?
  python - "$opts" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r03_n/lb8.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f"opts=[{sys.argv[1]}]", d['value'], round(d['ms_per_step']*1e3/d['config']['cg_iters_per_outer'],1), 'us/it', [(p['name'],round(p['avg_ms']*1e3,1)) for p in r['passes']])
except Exception as e: print(sys.argv[1], 'ERR', e, open('gpurun_out/r03_n/lb8.err').read()[-300:])
PY
  [ $rc -eq 124 ] && exit 1
done
echo finished
