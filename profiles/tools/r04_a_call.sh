#!/bin/bash
# round 4, call a: multi-process tests with the single-reduction CG + loopback A/B at 256^3 (8 slabs on one GPU), same box
OUT=gpurun_out/r04_a; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_multiproc.py -x -q --durations=10 > $OUT/pytest_multiproc.log 2>&1; echo "multiproc rc=$?"; tail -15 $OUT/pytest_multiproc.log
B="--steps 3 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5"
timeout -k 10 300 python bench.py $B > $OUT/bench_256_undivided.json 2> $OUT/und.err; echo "undivided rc=$?"
for o in 0 1; do
  NEUTFEM_OPTS="cg_single_reduce=$o" timeout -k 10 300 python bench.py --loopback-slabs 8 $B > $OUT/bench_256_lb8_cg1_$o.json 2> $OUT/lb8_$o.err; echo "lb8 cg1=$o rc=$?"
  NEUTFEM_OPTS="cg_single_reduce=$o" timeout -k 10 300 python bench.py --loopback-slabs 2 $B > $OUT/bench_256_lb2_cg1_$o.json 2> $OUT/lb2_$o.err; echo "lb2 cg1=$o rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_a/bench_256_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        cg = d["config"]["cg_iters_per_outer"]
        print(f"{f.split('/')[-1]:36s} value {d['value']:.4f} ms/step {d['ms_per_step']:.1f} cg/outer {cg} us/CG-it {1e3*d['ms_per_step']/cg:.1f} k {d['keff_after_timed_steps']:.13f}")
    except Exception as e:
        print(f, "unreadable", e)
PY
