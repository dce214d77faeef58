#!/bin/bash
# (1) is the 3.7e-9 difference of k between the one-rank and the two-rank 256^3 bench runs (CG tol 1e-4: 1047 vs 1050 iterations in the timed steps) a
#     stopping-threshold flip?  Same runs with CG tol 1e-10: the iterates must then agree to rounding.
# (2) the driver's launch line with 5 ranks on the one GPU (51/51/51/51/52 planes: unequal slabs -> scalar reduce route).  5 is the most the
#     box's process guard allows (the launcher itself holds the GPU open).
OUT=gpurun_out/r03_u; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-converge --no-parity --no-small --no-c5 --cpu-sample-iters 0 --cg-tol 1e-10 > $OUT/tight_n1.json 2> $OUT/tight_n1.err; rc=$?; echo "tight n1 rc=$rc"; [ $rc -eq 124 ] && exit 1
export NEUTFEM_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so NEUTFEM_FORCE_DEVICE=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 1 --warmup 1 --no-converge --cg-tol 1e-10 > $OUT/tight_n2.json 2> $OUT/tight_n2.err; rc=$?; echo "tight n2 rc=$rc"; [ $rc -eq 124 ] && exit 1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 5 --steps 2 --warmup 1 --no-converge > $OUT/bench_n5.json 2> $OUT/bench_n5.err; rc=$?; echo "n5 rc=$rc"; [ $rc -eq 124 ] && exit 1
python - <<'PY'
import json
for f in ("tight_n1","tight_n2","bench_n5"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03_u/{f}.json").read().splitlines() if l.startswith("{")][-1])
        print(f, d["n_gpus"], d["value"], repr(d["keff_after_timed_steps"]), d["config"]["cg_iters_per_outer"], d["config"]["parallelism"])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/r03_u/{f}.err").read()[-800:])
PY
echo finished
