#!/bin/bash
# the driver's N = 4 launch line at the full 256^3 size, four ranks on the ONE GPU of the box over the stand-in transport: per-rank slabs of
# 256 x 256 x 64, vector reduce, x || y, coarse start on the team in the converged solve -- k against the N = 1 line of the same box
OUT=gpurun_out/r03_s; mkdir -p $OUT
timeout -k 10 500 python bench.py --steps 2 --warmup 1 --cpu-sample-iters 0 --no-parity --no-small --no-c5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"
export NEUTFEM_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so NEUTFEM_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 2 --warmup 1 > $OUT/bench_n4.json 2> $OUT/bench_n4.err; echo "n4 rc=$?"
python - <<'PY'
import json
for f in ("bench_n1","bench_n4"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03_s/{f}.json").read().splitlines() if l.startswith("{")][-1])
        print(f, d["n_gpus"], d["value"], d["keff_after_timed_steps"], d.get("converged"))
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/r03_s/{f}.err").read()[-800:])
PY
echo finished
