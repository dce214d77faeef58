#!/bin/bash
# round 3, call b: long-line parity on the chunked kernel, A/B timings (512^3 x 2 groups, 256^3), PMC counter groups at 512^3 (classic vs chunked) and 128x128x512
OUT=gpurun_out/r03_b; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_longlines.py -q -x > $OUT/pytest_long.log 2>&1; rc=$?; echo "pytest long rc=$rc"; tail -4 $OUT/pytest_long.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
timeout -k 10 500 python profiles/tools/ab_long.py checker 512 2 10 > $OUT/ab_512.txt 2>&1; rc=$?; echo "ab 512 rc=$rc"; cat $OUT/ab_512.txt | cut -c1-200
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python profiles/tools/ab_long.py iaea3d 256 2 20 > $OUT/ab_256.txt 2>&1; rc=$?; echo "ab 256 rc=$rc"; cat $OUT/ab_256.txt | cut -c1-200
[ $rc -eq 124 ] && exit 1
bash profiles/tools/pmc_groups.sh r03_b_512_classic checker 512 512 512 2 4 s_long=0 || exit 1
python3 profiles/tools/pmc_summarize.py gpurun_out/pmc_r03_b_512_classic $OUT/pmc_512_classic.json "512^3 x 2 groups, classic one-chunk kernels (s_long=0)" > $OUT/pmc_512_classic_summary.txt 2>&1
bash profiles/tools/pmc_groups.sh r03_b_512_chunked checker 512 512 512 2 4 s_long=1 || exit 1
python3 profiles/tools/pmc_summarize.py gpurun_out/pmc_r03_b_512_chunked $OUT/pmc_512_chunked.json "512^3 x 2 groups, chunked long-line kernels (s_long=1)" > $OUT/pmc_512_chunked_summary.txt 2>&1
bash profiles/tools/pmc_groups.sh r03_b_128x128x512 uniform 128 128 512 1 4 s_long=0 || exit 1
python3 profiles/tools/pmc_summarize.py gpurun_out/pmc_r03_b_128x128x512 $OUT/pmc_128x128x512_classic.json "128x128x512, classic kernels" > $OUT/pmc_128_summary.txt 2>&1
rm -rf gpurun_out/pmc_r03_b_*/g*/ 2>/dev/null
echo finished
