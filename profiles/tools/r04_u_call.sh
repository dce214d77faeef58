#!/bin/bash
# round 4, call u: the y passes of x || y on the comm stream behind the plane exchange (one fork / join pair less per apply) against a stream of their own
OUT=gpurun_out/r04_u; mkdir -p $OUT
B="--steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5"
run() {
  NEUTFEM_HIP_LIB="$PWD/scratch/libs/libnf_ycomm.so" NEUTFEM_OPTS="$2" timeout -k 10 300 python bench.py $B $3 > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$1" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_u/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    print(f"{sys.argv[1]:40s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:7.1f}  k {d['keff_after_timed_steps']:.12f}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
}
for lb in 8 4; do
run "$lb slabs, y on its own stream" "y_on_comm=0" "--loopback-slabs $lb"
run "$lb slabs, y behind the exchange" "y_on_comm=1" "--loopback-slabs $lb"
run "$lb slabs, y on its own stream" "y_on_comm=0" "--loopback-slabs $lb"
run "$lb slabs, y behind the exchange" "y_on_comm=1" "--loopback-slabs $lb"
done
