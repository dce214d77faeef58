#!/bin/bash
# round 4, call o: do the small slabs want the streaming-load instantiations or the XCD-contiguous tile order?  8-slab loopback at 256^3, default CG route
OUT=gpurun_out/r04_o; mkdir -p $OUT
B="--steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 --loopback-slabs 8"
for opts in "" "nt_min_cells=0" "xcd=1" "xcd=2" "xcd=3" "nt_min_cells=0,xcd=3" "xy_overlap=0"; do
  NEUTFEM_OPTS="$opts" timeout -k 10 300 python bench.py $B > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$opts" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_o/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.1f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1] or 'default':28s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:7.1f}  passes(us) {ps}  k {d['keff_after_timed_steps']:.12f}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
done
