#!/bin/bash
# round 4, call r: the accumulation pass of the single-reduction CG compiled for 3 waves per SIMD (141 VGPRs, no scratch) against the default 4 (128 VGPRs, 60 B of
# scratch) on large slabs: 512^3 x 2 groups as 8 slabs, 256^3 as 2 and 4 slabs
OUT=gpurun_out/r04_r; mkdir -p $OUT
run() {
  NEUTFEM_HIP_LIB="$2" timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 $3 > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$1" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_r/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.0f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1]:44s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:8.1f}  passes(us) {ps}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
}
W3="$PWD/scratch/libs/libnf_z2w3.so"
C="--case checker --n 512 --groups 2"
run "512^3 8 slabs, 4 waves (default)" "" "$C --loopback-slabs 8"
run "512^3 8 slabs, 3 waves" "$W3" "$C --loopback-slabs 8"
run "256^3 2 slabs, 4 waves (default)" "" "--loopback-slabs 2"
run "256^3 2 slabs, 3 waves" "$W3" "--loopback-slabs 2"
run "256^3 4 slabs, 4 waves (default)" "" "--loopback-slabs 4"
run "256^3 4 slabs, 3 waves" "$W3" "--loopback-slabs 4"
run "256^3 8 slabs, 4 waves (default)" "" "--loopback-slabs 8"
run "256^3 8 slabs, 3 waves" "$W3" "--loopback-slabs 8"
