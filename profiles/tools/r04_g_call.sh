#!/bin/bash
OUT=gpurun_out/r04_g; mkdir -p $OUT
timeout -k 10 600 python profiles/tools/r04_g_higher_orders.py > $OUT/higher_orders.txt 2>&1; echo "rc=$?"; cat $OUT/higher_orders.txt | cut -c1-400
NEUTFEM_COPY_VERBOSE=1 timeout -k 10 120 python - <<'PY' 2>&1 | tail -20
import sys
sys.path.insert(0, "tests")
from helpers import make_hip, synthetic_inputs
s = make_hip(synthetic_inputs(8, 8, 8, 1))
print("best", s.time_device_copy(1 << 30, 20))
PY
