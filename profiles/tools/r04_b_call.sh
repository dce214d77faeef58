#!/bin/bash
# round 4, call b: per-kernel durations of the 8-slab loopback at 256^3, two-reduction vs single-reduction CG (rocprofv3 --kernel-trace --stats)
OUT=gpurun_out/r04_b; mkdir -p $OUT; export TMPDIR=/tmp
B="--steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 --loopback-slabs 8"
for o in 0 1; do
  NEUTFEM_OPTS="cg_single_reduce=$o" timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cg1_$o -o st -- python3 bench.py $B > $OUT/bench_cg1_$o.json 2> $OUT/cg1_$o.err; echo "cg1=$o rc=$?"
  f=$(find $OUT/cg1_$o -name "*kernel_stats.csv" | head -1); echo "== cg_single_reduce=$o: $f"; python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f"{re.sub(r'^void nf::','',r['Name'])[:95]:95s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us total {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f} %")
PY
done
