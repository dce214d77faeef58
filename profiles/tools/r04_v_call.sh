#!/bin/bash
# round 4, call v: options of the chunked long-line passes at 512^3 x 2 groups (undivided), fixed work
OUT=gpurun_out/r04_v; mkdir -p $OUT
B="--case checker --n 512 --groups 2 --steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5"
for opts in "" "xcd=2" "xcd=3" "xcd=0" "s_tx=16" "s_tx=64" ""; do
  NEUTFEM_OPTS="$opts" timeout -k 10 300 python bench.py $B > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$opts" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_v/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.0f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1] or 'default':12s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:8.1f}  passes(us) {ps}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
done
