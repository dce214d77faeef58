#!/bin/bash
# the driver's launch line at 256^3 with 2 and 6 ranks on the ONE GPU of the box over the stand-in transport (6 = 43 / 43 / 42 / 43 / 43 / 42 planes:
# unequal slabs -> scalar reduce route; 2 = 8.4 M-cell slabs with streaming loads)
OUT=gpurun_out/r03_t; mkdir -p $OUT
export NEUTFEM_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so NEUTFEM_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 6; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 2 --warmup 1 --no-converge > $OUT/bench_n$n.json 2> $OUT/bench_n$n.err; rc=$?; echo "n$n rc=$rc"
  [ $rc -eq 124 ] && exit 1
done
python - <<'PY'
import json
for n in (2,6):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03_t/bench_n{n}.json").read().splitlines() if l.startswith("{")][-1])
        print(n, d["n_gpus"], d["value"], repr(d["keff_after_timed_steps"]), d["config"]["parallelism"])
    except Exception as e: print(n, "ERR", e, open(f"gpurun_out/r03_t/bench_n{n}.err").read()[-800:])
PY
echo finished
