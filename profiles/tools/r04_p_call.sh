#!/bin/bash
# round 4, call p: is the power-of-two plane stride (256 x 256 x 8 B = 512 KiB) what holds the slab passes back?  the same 8-slab loopback on meshes around 256^3
OUT=gpurun_out/r04_p; mkdir -p $OUT
B="--steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5"
for n in 256 248 264 240 272; do
 for lb in 1 8; do
  NEUTFEM_BENCH_N=$n timeout -k 10 300 python bench.py $B --loopback-slabs $lb > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$n" $lb $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_p/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]; N = d["config"]["cells"]
    us = 1e3*d['ms_per_step']/cg
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.1f}" for p in d["roofline"]["passes"])
    print(f"n={sys.argv[1]:4s} slabs={sys.argv[2]} rc {sys.argv[3]} us/CG-it {us:7.1f}  ps per cell and CG-it {1e6*us/N:6.2f}  passes(us) {ps}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "rc", sys.argv[3], "unreadable", e)
PY
 done
done
