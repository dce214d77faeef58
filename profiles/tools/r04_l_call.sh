#!/bin/bash
# round 4, call l: where does the single-reduction CG with the weighted-sum endpoint pass stop paying?  512^3 x 2 groups as 8 slabs (16.8 M cells each) and
# 256^3 as 2 and 4 slabs (8.4 M / 4.2 M cells each), forced on and off
OUT=gpurun_out/r04_l; mkdir -p $OUT
run() {
  NEUTFEM_OPTS="$2" timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 $3 > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$1" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_l/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.0f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1]:58s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:8.1f}  passes(us) {ps}  k {d['keff_after_timed_steps']:.12f}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
}
C="--case checker --n 512 --groups 2"
run "512^3 8 slabs, two reductions" "cg_single_reduce=0" "$C --loopback-slabs 8"
run "512^3 8 slabs, single reduction + weighted endpoint" "cg_single_reduce=1" "$C --loopback-slabs 8"
run "256^3 2 slabs, two reductions" "cg_single_reduce=0" "--loopback-slabs 2"
run "256^3 2 slabs, single reduction + weighted endpoint" "cg_single_reduce=1" "--loopback-slabs 2"
run "256^3 4 slabs, two reductions" "cg_single_reduce=0" "--loopback-slabs 4"
run "256^3 4 slabs, single reduction + weighted endpoint" "cg_single_reduce=1" "--loopback-slabs 4"
