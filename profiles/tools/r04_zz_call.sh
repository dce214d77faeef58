#!/bin/bash
# round 4, call h: whole GPU suite with durations, the default bench line, rocprofv3 evidence (stats + PMC passes) of the same commit
OUT=gpurun_out/r04_zz; mkdir -p $OUT
t0=$(date +%s); timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=40 > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -4 $OUT/pytest_gpu_full.log
t0=$(date +%s); timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
NEUTFEM_COMMIT=fd3e415 timeout -k 10 900 bash profiles/collect.sh r04_zz > $OUT/collect.log 2>&1; echo "collect rc=$?"; tail -12 $OUT/collect.log
cp gpurun_out/prof_r04_zz/r04_zz_* $OUT/ 2>/dev/null; cp gpurun_out/prof_r04_zz/bench_stats.json $OUT/bench_under_rocprof.json 2>/dev/null
rm -rf gpurun_out/prof_r04_zz/stats gpurun_out/prof_r04_zz/fetch gpurun_out/prof_r04_zz/write
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_zz/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "avg_ms")})
print("passes", d["roofline"]["passes"], "copy", d["roofline"]["measured_copy"])
print("higher_order", d.get("higher_order")); print("c5", {k: v for k, v in d.get("c5_single_gpu", {}).items() if k != "parity"}); print("c5 parity", d.get("c5_single_gpu", {}).get("parity"))
print("converged", d.get("converged")); print("parity", d.get("parity"), d.get("parity_at_bench_size"))
for c in d.get("other_configs", []): print(c["config"][:60], c["solve_ms"], c["path"], c["pcm_vs_oracle"], c["cg_iterations"], c["cg_iterations_oracle"])
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind")})
PY
for lb in 8 2; do timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample-iters 0 --no-parity --no-small --no-c5 --loopback-slabs $lb > gpurun_out/r04_zz/bench_256cube_loopback$lb.json 2> gpurun_out/r04_zz/lb$lb.err; echo "loopback $lb rc=$?"; done
