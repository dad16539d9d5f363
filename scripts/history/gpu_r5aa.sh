#!/bin/bash
# round 4, call AA: GroupNorm apply writes the token-major operand image of a Linear proj_in (SDXL SpatialTransformer): op test, SDXL model tests, family A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "group_norm or concat" ) > gpurun_out/r5aa_tests_ops.log 2>&1; echo "rc=$?" >> gpurun_out/r5aa_tests_ops.log
( timeout 400 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "sdxl or unet" ) > gpurun_out/r5aa_tests_models.log 2>&1; echo "rc=$?" >> gpurun_out/r5aa_tests_models.log
rm -f gpurun_out/r5aa_family_sdxl.txt
for rep in 1 2; do for o in "fuse_gn_tokens=0" "fuse_gn_tokens=1"; do echo "#### sdxl $o" >> gpurun_out/r5aa_family_sdxl.txt; timeout 300 python scripts/family_times.py sdxl $o 2>&1 | head -18 >> gpurun_out/r5aa_family_sdxl.txt; done; done
tail -n 3 gpurun_out/r5aa_tests_ops.log gpurun_out/r5aa_tests_models.log; grep "####\|==\|f32 norms\|copies\|f32 rows\|GroupNorm" gpurun_out/r5aa_family_sdxl.txt
