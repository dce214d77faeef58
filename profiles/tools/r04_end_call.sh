#!/bin/bash
# round 4, last call: the whole GPU suite (with durations), smoke() and the default bench line on the final commit
OUT=gpurun_out/r04_end2; mkdir -p $OUT
python -m pytest tests -x -q -m gpu --durations=25 > $OUT/pytest_gpu_full.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $OUT/pytest_gpu_full.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_end2/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "avg_ms")}, d["roofline"].get("traffic_source"))
PY
