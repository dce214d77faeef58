#!/bin/bash
# kernel trace of one IAEA-3D 38x38x19 solve as the driver runs it (coarse start): where does the time outside k_keff_xcd go?
OUT=$PWD/gpurun_out/r03_x; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $REPO/profiles/tools/r02/small_trace.py iaea3d 0 1 > $OUT/trace.log 2>&1; rc=$?; tail -3 $OUT/trace.log; echo "rc=$rc"
cd $REPO; python3 profiles/tools/r02/trace_gaps.py $OUT/trace > $OUT/trace_gaps.txt 2>&1; cat $OUT/trace_gaps.txt
