#!/bin/bash
# round 4, call f: the tests this round added or changed + the copy microbenchmark variants
OUT=gpurun_out/r04_f; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_parity.py tests/test_gpu_longlines.py tests/test_gpu_slabs.py -x -q --durations=25 -k "multiproc or 256cube or 512cube or team_solve_keff or thin_slabs" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest.log
NEUTFEM_COPY_VERBOSE=1 timeout -k 10 120 python - <<'PY' 2>&1 | tail -16
import numpy as np, sys
sys.path.insert(0, "tests")
from helpers import make_hip, synthetic_inputs
s = make_hip(synthetic_inputs(8, 8, 8, 1))
print("best", s.time_device_copy(1 << 30, 20))
PY
