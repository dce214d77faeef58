#!/bin/bash
# round 4, call q: the x pass with DPP moves instead of ds_bpermute in its scans (scratch/libs/libnf_xdpp.so, option x_dpp): 8-slab loopback at 256^3 and the undivided mesh
OUT=gpurun_out/r04_q; mkdir -p $OUT
B="--steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5"
run() {
  NEUTFEM_HIP_LIB="$PWD/scratch/libs/libnf_xdpp.so" NEUTFEM_OPTS="$2" timeout -k 10 300 python bench.py $B $3 > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$1" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_q/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.1f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1]:34s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:7.1f}  passes(us) {ps}  k {d['keff_after_timed_steps']:.12f}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
}
run "8 slabs, crossbar scans" "x_dpp=0" "--loopback-slabs 8"
run "8 slabs, DPP scans" "x_dpp=1" "--loopback-slabs 8"
run "8 slabs, crossbar scans" "x_dpp=0" "--loopback-slabs 8"
run "8 slabs, DPP scans" "x_dpp=1" "--loopback-slabs 8"
run "4 slabs, crossbar scans" "x_dpp=0" "--loopback-slabs 4"
run "4 slabs, DPP scans" "x_dpp=1" "--loopback-slabs 4"
