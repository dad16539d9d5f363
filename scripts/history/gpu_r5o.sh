#!/bin/bash
# round 4, call O: GEGLU FF1 on any tile (16-column interleave, epi_geglu16): op tests, model tests, SDXL / SD1.5 family A/B (geglu16 0 / 1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "geglu or feed_forward or unet or sdxl" ) > gpurun_out/r5o_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5o_tests.log
rm -f gpurun_out/r5o_family.txt
for rep in 1 2; do for o in "geglu16=0" "geglu16=1"; do for m in sdxl sd15; do echo "#### $m $o" >> gpurun_out/r5o_family.txt; timeout 300 python scripts/family_times.py $m $o 2>&1 | head -4 >> gpurun_out/r5o_family.txt; done; done; done
timeout 200 python scripts/ab_bench.py geglu16 0,1 3 4 > gpurun_out/r5o_ab_sd15.txt 2>&1
tail -n 4 gpurun_out/r5o_tests.log; grep "####\|==\|Linear" gpurun_out/r5o_family.txt; tail -n 2 gpurun_out/r5o_ab_sd15.txt
