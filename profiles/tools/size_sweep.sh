#!/bin/bash
# outer iterations/s of the IAEA-3D workload resampled to n^3 (2 groups), one GPU: python bench.py with the extras switched off
OUT=${1:-gpurun_out/sweep}; mkdir -p $OUT
for n in 64 96 128 192 256 384 512; do
  NEUTFEM_BENCH_N=$n timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  rc=$?; echo "n=$n rc=$rc"; [ $rc -eq 124 ] && exit 1
done
python - "$OUT" <<'PY'
import json, sys, glob, os
print("| n | outer-iters/s | CG its / outer | us per CG iteration | x / y / z pass us (HIP events) | algorithmic TB/s of the passes |")
print("|---|---|---|---|---|---|")
for n in (64, 96, 128, 192, 256, 384, 512):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], f"bench_{n}.json")).read().strip().splitlines()[-1])
    except Exception as e:
        print(f"| {n} | error {e} |"); continue
    r = d["roofline"]; cg = d["config"]["cg_iters_per_outer"]
    ps = {p["name"]: p for p in r["passes"]}
    t = " / ".join(f"{ps[k]['avg_ms'] * 1e3:.1f}" for k in ps); a = " / ".join(f"{ps[k]['achieved'] / 1e3:.2f}" for k in ps)
    print(f"| {n} | {d['value']:.3f} | {cg:.0f} | {d['ms_per_step'] * 1e3 / cg:.1f} | {t} | {a} |")
PY
