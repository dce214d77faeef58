#!/bin/bash
# round 4: bench.py --gpus 2 / 4 at the REAL bench size (IAEA-3D 256^3), all ranks on the one GPU of the box over the stand-in transport (tests/fake_rccl):
# the code path the driver's N = 2 / 4 / 8 runs take (bench.py starts its own ranks, gloo rendezvous, slab split, middle ranks with two neighbours, barrier +
# max-over-ranks timing), rehearsed at full message and slab sizes.  The rates are NOT scaling numbers (one GPU, host-staged transport); k-eff is the evidence.
OUT=gpurun_out/r04_rehearsal; mkdir -p $OUT
export NEUTFEM_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so NEUTFEM_FORCE_DEVICE=0 FAKE_RCCL_REPORT=1
for n in 2 4; do
  timeout -k 10 400 python bench.py --gpus $n --steps 2 --warmup 1 > $OUT/bench_gpus$n.json 2> $OUT/bench_gpus$n.err; rc=$?; echo "gpus=$n rc=$rc"
  [ $rc -eq 0 ] || { tail -20 $OUT/bench_gpus$n.err; exit $rc; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_gpus$n.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("n_gpus", "rccl_ranks", "transport", "value", "ms_per_step", "scaling", "keff_after_timed_steps")}, d.get("converged"), d.get("config"))
PY
  grep "fake_rccl: rank" $OUT/bench_gpus$n.err | head -8
done
