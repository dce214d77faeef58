#!/bin/bash
# round 4, call k: the endpoint pass as weighted sums (k_endpoint_w): parity tests, then A/B in the 8-slab loopback at 256^3 + per-kernel durations
OUT=gpurun_out/r04_k; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_multiproc.py tests/test_gpu_parity.py -x -q --durations=8 -k "slabs or multiproc or 256cube_golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest.log
B="--steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 --loopback-slabs 8"
run() {
  NEUTFEM_OPTS="$2" timeout -k 10 300 python bench.py $B > $OUT/b.json 2> $OUT/b.err; rc=$?
  python - "$1" $rc <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04_k/b.json").read().strip().splitlines()[-1]); cg = d["config"]["cg_iters_per_outer"]
    ps = " ".join(f"{p['name'][-1]} {1e3*p['avg_ms']:.1f}" for p in d["roofline"]["passes"])
    print(f"{sys.argv[1]:52s} rc {sys.argv[2]} us/CG-it {1e3*d['ms_per_step']/cg:7.1f}  passes(us) {ps}  k {d['keff_after_timed_steps']:.12f}")
except Exception as e:
    print(sys.argv[1], "rc", sys.argv[2], "unreadable", e)
PY
}
run "two reductions" "cg_single_reduce=0"
run "single reduction, chain-solve endpoint pass" "cg_single_reduce=1,endpoint_weights=0"
run "single reduction, weighted-sum endpoint pass" "cg_single_reduce=1,endpoint_weights=1"
NEUTFEM_OPTS="cg_single_reduce=1,endpoint_weights=1" timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o st -- python3 bench.py $B > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f"{re.sub(r'^void nf::','',r['Name'])[:80]:80s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.1f} %")
PY
rm -f $OUT/prof/*kernel_trace.csv $OUT/prof/*/*kernel_trace.csv
