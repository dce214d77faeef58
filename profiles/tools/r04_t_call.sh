#!/bin/bash
# round 4, call t: how much of the 8-slab loopback iteration is idle time between kernels?  (rocprofv3 kernel trace: union of busy intervals against wall time)
OUT=gpurun_out/r04_t; mkdir -p $OUT; export TMPDIR=/tmp
B="--steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 --loopback-slabs 8"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o tr -- python3 bench.py $B > $OUT/bench.json 2> $OUT/err.txt; echo "rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys, re
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the timed region: the last 45 % of the trace (two timed outers of three) -- take kernels named k_schur / k_endpoint / k_reduce window
core = [r for r in rows if re.search(r"k_schur_|k_endpoint_w|k_reduce_rows|copyBuffer", r[2])]
n = len(core); core = core[n // 2:]                      # second half: steady state
t0, t1 = core[0][0], max(r[1] for r in core)
busy = 0; cur_s, cur_e = core[0][0], core[0][1]
for s, e, _ in core[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
its = sum(1 for r in core if "k_reduce_rows" in r[2])
print(f"steady-state window {(t1 - t0) / 1e3:.0f} us, {its} CG iterations: {(t1 - t0) / 1e3 / its:.1f} us per iteration, GPU busy (union of kernels) {100.0 * busy / (t1 - t0):.1f} %, idle {(t1 - t0 - busy) / 1e3 / its:.1f} us per iteration")
gaps = []
prev_e = core[0][1]
for s, e, _ in core[1:]:
    if s > prev_e: gaps.append(s - prev_e)
    prev_e = max(prev_e, e)
gaps.sort()
print(f"gaps: n {len(gaps)} median {gaps[len(gaps)//2] / 1e3:.2f} us, p90 {gaps[int(len(gaps)*0.9)] / 1e3:.2f} us, per iteration {len(gaps) / its:.1f}")
PY
rm -rf $OUT/prof
