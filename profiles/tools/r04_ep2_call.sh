#!/bin/bash
# round 4, EXPERIMENT (apply profiles/tools/r04_endpoint_w2.patch and rebuild first; results: profiles/r04_y_endpoint_pass_probes.txt; not adopted)
# endpoint pass with two columns per lane (k_endpoint_w2): parity of the slab tests, then 8-slab loopback at 256^3 with its variants in one call:
# endpoint_weights = 1 straight-line whole tiles (default), 2 one column per lane (k_endpoint_w), 3 pairs, one plane after the other, 4 pairs, staged + predicated
OUT=gpurun_out/r04_ep2; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_slabs.py -x -q -k "endpoint or team_solve_keff or schur_apply" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
for w in 1 2 3 4 1 2 3 4; do
  NEUTFEM_OPTS=endpoint_weights=$w timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-sample-iters 0 --no-parity --no-small --no-c5 --no-converge --loopback-slabs 8 > $OUT/lb8_w$w.json 2> $OUT/lb8_w$w.err || exit 1
  python - <<PY
import json
d = json.loads(open("$OUT/lb8_w$w.json").read().strip().splitlines()[-1])
it = d.get("config", {}).get("cg_iters_per_outer", 0)
print("endpoint_weights=$w", "ms/step", d["ms_per_step"], "us per CG iteration", round(1e3 * d["ms_per_step"] / it, 1) if it else None, "keff", d.get("keff_after_timed_steps"))
PY
done
