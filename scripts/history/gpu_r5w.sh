#!/bin/bash
# round 4, call W: skip-connection CONCAT never materialised (plan_concat_gn): op test, UNet model tests, full-width tests, pixels test, SD1.5 / SDXL A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "concat or group_norm or conv2d" ) > gpurun_out/r5w_tests_ops.log 2>&1; echo "rc=$?" >> gpurun_out/r5w_tests_ops.log
( timeout 600 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py tests/test_zz_gpu_pixels.py -m gpu -q -x -k "unet or sdxl or sd15 or pixel or sampler or generate" ) > gpurun_out/r5w_tests_models.log 2>&1; echo "rc=$?" >> gpurun_out/r5w_tests_models.log
timeout 200 python scripts/ab_bench.py fuse_concat_gn 0,1 3 4 > gpurun_out/r5w_ab_sd15.txt 2>&1
rm -f gpurun_out/r5w_family.txt
for rep in 1 2; do for o in "fuse_concat_gn=0" "fuse_concat_gn=1"; do for m in sd15 sdxl; do echo "#### $m $o" >> gpurun_out/r5w_family.txt; timeout 300 python scripts/family_times.py $m $o 2>&1 | head -16 >> gpurun_out/r5w_family.txt; done; done; done
tail -n 3 gpurun_out/r5w_tests_ops.log gpurun_out/r5w_tests_models.log gpurun_out/r5w_ab_sd15.txt; grep "####\|==\|concat\|GroupNorm" gpurun_out/r5w_family.txt
