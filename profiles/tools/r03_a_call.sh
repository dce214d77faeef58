#!/bin/bash
# round 3, call a: access-pattern microbenchmark at 512^3, PMC counters of the round-2 y/z kernels at 512^3 and 128x128x512, long-line parity tests
OUT=gpurun_out/r03_a; mkdir -p $OUT
timeout -k 10 300 profiles/tools/tile_copy512 512 > $OUT/tile_copy512.txt 2>&1 || { echo "tile_copy failed rc=$?"; tail -5 $OUT/tile_copy512.txt; exit 1; }
echo "tile copy done"; grep -c TB $OUT/tile_copy512.txt
timeout -k 10 900 python -m pytest tests/test_gpu_longlines.py -q -x -k "test_schur_apply_long_lines or test_solve_keff_c5" > $OUT/pytest_long.log 2>&1; echo "pytest long rc=$?"; tail -5 $OUT/pytest_long.log | cut -c1-400
bash profiles/tools/pmc_groups.sh r03_a_512 checker 512 512 512 2 4 || exit 1
bash profiles/tools/pmc_groups.sh r03_a_128x128x512 uniform 128 128 512 1 4 || exit 1
python3 profiles/tools/pmc_summarize.py gpurun_out/pmc_r03_a_512 $OUT/pmc_512.json > $OUT/pmc_512_summary.txt 2>&1
python3 profiles/tools/pmc_summarize.py gpurun_out/pmc_r03_a_128x128x512 $OUT/pmc_128x128x512.json > $OUT/pmc_128_summary.txt 2>&1
rm -rf gpurun_out/pmc_r03_a_512/g*/ gpurun_out/pmc_r03_a_128x128x512/g*/ 2>/dev/null
echo finished
