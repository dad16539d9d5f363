#!/bin/bash
# round 4, call C: fused split-K reduce + LayerNorm, one row per wave
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "split_linear_reduce or layer_norm or linear or feed_forward or spatial" ) > gpurun_out/r5c_tests_default.log 2>&1; echo "rc=$?" >> gpurun_out/r5c_tests_default.log
( timeout 300 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "unet or sdxl" ) > gpurun_out/r5c_tests_models.log 2>&1; echo "rc=$?" >> gpurun_out/r5c_tests_models.log
rm -f gpurun_out/r5c_family_sdxl.txt
for o in "fuse_ln_reduce=0" "fuse_ln_reduce=1" "fuse_ln_reduce=0" "fuse_ln_reduce=1"; do
  echo "#### sdxl $o" >> gpurun_out/r5c_family_sdxl.txt
  timeout 200 python scripts/family_times.py sdxl $o 2>&1 | head -12 >> gpurun_out/r5c_family_sdxl.txt
done
timeout 200 python scripts/ab_bench.py fuse_ln_reduce 0,1 3 4 > gpurun_out/r5c_ab_fuse_ln_reduce.txt 2>&1
tail -n 4 gpurun_out/r5c_tests_*.log; grep "####\|==\|split-K\|LayerNorm" gpurun_out/r5c_family_sdxl.txt; tail -n 3 gpurun_out/r5c_ab_*.txt
