#!/bin/bash
# (1) resident one-workgroup kernels before / after the per-direction pointer tables left scratch memory (same box, two builds)
# (2) k after 3 outers with CG tol 1e-10 on 1, 2 and 3 ranks (256^3, stand-in transport): the iterates must agree to rounding
OUT=gpurun_out/r03_v; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NEUTFEM_HIP_LIB=$PWD/profiles/tools/_ab/libneutfem_hip_prev.so timeout -k 10 200 python profiles/tools/ab_small.py 15 > $OUT/ab_small_prev.txt 2>&1 || { echo "prev failed"; tail -5 $OUT/ab_small_prev.txt; exit 1; }
timeout -k 10 200 python profiles/tools/ab_small.py 15 > $OUT/ab_small_new.txt 2>&1 || { echo "new failed"; tail -5 $OUT/ab_small_new.txt; exit 1; }
NEUTFEM_HIP_LIB=$PWD/profiles/tools/_ab/libneutfem_hip_prev.so timeout -k 10 200 python profiles/tools/ab_small.py 15 > $OUT/ab_small_prev2.txt 2>&1
timeout -k 10 200 python profiles/tools/ab_small.py 15 > $OUT/ab_small_new2.txt 2>&1
cat $OUT/ab_small_prev.txt $OUT/ab_small_new.txt $OUT/ab_small_prev2.txt $OUT/ab_small_new2.txt
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q -m gpu > $OUT/pytest.txt 2>&1; rc=$?; tail -3 $OUT/pytest.txt; [ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 0 --no-converge --no-parity --no-small --no-c5 --cpu-sample-iters 0 --cg-tol 1e-10 > $OUT/tight_n1.json 2> $OUT/tight_n1.err; rc=$?; echo "tight n1 rc=$rc"; [ $rc -eq 124 ] && exit 1
export NEUTFEM_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so NEUTFEM_FORCE_DEVICE=0
for n in 2 3; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --steps 3 --warmup 0 --no-converge --cg-tol 1e-10 > $OUT/tight_n$n.json 2> $OUT/tight_n$n.err; rc=$?; echo "tight n$n rc=$rc"; [ $rc -eq 124 ] && exit 1
done
python - <<'PY'
import json
for f in ("tight_n1","tight_n2","tight_n3"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03_v/{f}.json").read().splitlines() if l.startswith("{")][-1])
        print(f, d["n_gpus"], d["value"], repr(d["keff_after_timed_steps"]), d["config"]["cg_iters_per_outer"], d["config"]["parallelism"])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/r03_v/{f}.err").read()[-800:])
PY
echo finished
