#!/bin/bash
# round 4, call j: 512^3 x 2 groups as 8 slabs of 512 x 512 x 64 on one GPU (loopback), both CG routes, against the undivided mesh (fixed work: 50 CG iterations per solve)
OUT=gpurun_out/r04_j; mkdir -p $OUT
B="--case checker --n 512 --groups 2 --steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5"
run() {
  NEUTFEM_OPTS="$2" timeout -k 10 400 python bench.py $B $3 > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$1" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_j/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.0f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1]:44s} rc {sys.argv[2]} ms/CG-it {d['ms_per_step']/cg:7.3f}  passes(us) {ps}  k {d['keff_after_timed_steps']:.12f}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
}
run "undivided 512^3 x 2 groups" "" ""
run "8 slabs, two reductions" "cg_single_reduce=0" "--loopback-slabs 8"
run "8 slabs, single reduction" "cg_single_reduce=1" "--loopback-slabs 8"
