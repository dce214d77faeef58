#!/bin/bash
# round 3, call o: x || y on small slabs -- slab / multi-process tests, then the 8- and 2-slab loopback with the overlap on / off
OUT=gpurun_out/r03_o; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_multiproc.py -q -x > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest.log | cut -c1-250
[ $rc -eq 124 ] && exit 1
for lb in 8 2; do for opts in "xy_overlap=1" "xy_overlap=0"; do
  NEUTFEM_OPTS="$opts" timeout -k 10 300 python bench.py --loopback-slabs $lb --steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/lb.json 2> $OUT/lb.err; rc=$?
  python - "$lb slabs, $opts" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r03_o/lb.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f"[{sys.argv[1]}]", d['value'], round(d['ms_per_step']*1e3/d['config']['cg_iters_per_outer'],1), 'us/it', d['keff_after_timed_steps'], [(p['name'],round(p['avg_ms']*1e3,1)) for p in r['passes']])
except Exception as e: print(sys.argv[1], 'ERR', e, open('gpurun_out/r03_o/lb.err').read()[-300:])
PY
  [ $rc -eq 124 ] && exit 1
done; done
echo finished
