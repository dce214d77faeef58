#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box from the repo root):
#   1. --kernel-trace --stats         -> per-kernel average durations (must agree with bench.py's HIP-event timings)
#   2. --pmc FETCH_SIZE (own pass)    -> HBM read traffic per dispatch
#   3. --pmc WRITE_SIZE (own pass)    -> HBM write traffic per dispatch
# then profiles/summarize.py condenses the CSVs into profiles/<tag>_kernel_stats.csv and <tag>_pmc_traffic.json.
set -e
TAG=${1:-r03_zzzz}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
ARGS="--steps 2 --warmup 1 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $REPO/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample-iters 0 --no-converge --no-parity --no-small --no-c5 > $OUT/bench_write.json 2> $OUT/write.err
echo "write pass done"
cd $REPO
python3 profiles/summarize.py $OUT $TAG
