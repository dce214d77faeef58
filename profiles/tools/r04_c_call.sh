#!/bin/bash
# round 4, call c: tile shape of the slab passes at strong-scaling size (256 x 256 x 32 slabs, 8-slab loopback), both CG routes
OUT=gpurun_out/r04_c; mkdir -p $OUT
B="--steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 --loopback-slabs 8"
for opts in "cg_single_reduce=0" "cg_single_reduce=0,s_tx=32" "cg_single_reduce=0,s_tx=16" "cg_single_reduce=0,s_seg=4" "cg_single_reduce=0,s_seg=4,s_tx=32" "cg_single_reduce=1" "cg_single_reduce=1,s_tx=32" "cg_single_reduce=1,s_tx=16"; do
  NEUTFEM_OPTS="$opts" timeout -k 10 300 python bench.py $B > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$opts" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_c/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.1f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1]:42s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:7.1f}  passes(us) {ps}  k {d['keff_after_timed_steps']:.12f}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
done
