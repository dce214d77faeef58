#!/bin/bash
# round 4, call n: the accumulation pass with its three block sums behind one pair of barriers (block_sum3): parity tests + 8-slab loopback at 256^3
OUT=gpurun_out/r04_n; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_parity.py -x -q -k "slabs or 256cube_golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="--steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 --loopback-slabs 8"
for i in 1 2; do
timeout -k 10 300 python bench.py $B > $OUT/b.json 2> $OUT/b.err; rc=$?
python - $rc <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_n/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.1f}" for p in d["roofline"]["passes"])
print(f"default (single reduction, weighted endpoint, block_sum3) rc {sys.argv[1]} us/CG-it {1e3*d['ms_per_step']/cg:7.1f}  passes(us) {ps}  k {d['keff_after_timed_steps']:.12f}")
PY
done
