#!/bin/bash
# round 4, call zzz: multi-process + slab tests and the rocprofv3 evidence (stats + PMC passes) of the final commit
OUT=gpurun_out/r04_zzz; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_slabs.py tests/test_gpu_parity.py -q -k "multiproc or slabs or 256cube" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
NEUTFEM_COMMIT=cb2435e timeout -k 10 900 bash profiles/collect.sh r04_zzz > $OUT/collect.log 2>&1; echo "collect rc=$?"; tail -4 $OUT/collect.log
cp gpurun_out/prof_r04_zzz/r04_zzz_* $OUT/ 2>/dev/null; cp gpurun_out/prof_r04_zzz/bench_stats.json $OUT/bench_under_rocprof.json 2>/dev/null
rm -rf gpurun_out/prof_r04_zzz/stats gpurun_out/prof_r04_zzz/fetch gpurun_out/prof_r04_zzz/write
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_zzz/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "avg_ms")}, d["roofline"].get("traffic_source"))
PY
