#!/bin/bash
# the new XCD-path tests, then evidence of the final kernels: rocprofv3 kernel stats + PMC traffic of the headline (profiles/collect.sh), then the default bench line
OUT=gpurun_out/r03_zzzz; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo skip

timeout -k 10 700 bash profiles/collect.sh r03_zzzz > $OUT/collect.log 2>&1; rc=$?; tail -5 $OUT/collect.log; echo "collect rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; echo "bench rc=$rc"; tail -c 600 $OUT/bench.json
