#!/bin/bash
# round 3, call N: concat-along-features -> operand image (FLUX single block tail): tests + FLUX family times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "single_block_tail or flux or Flux or FLUX or dit or mmdit" ) > gpurun_out/r3n_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3n_tests.log
( timeout 600 python scripts/family_times.py flux ) > gpurun_out/r3n_family_flux.txt 2>&1
( timeout 600 python scripts/family_times.py flux fuse_cat_rows16=0 ) > gpurun_out/r3n_family_flux_off.txt 2>&1
tail -5 gpurun_out/r3n_tests.log; head -16 gpurun_out/r3n_family_flux.txt; head -3 gpurun_out/r3n_family_flux_off.txt
