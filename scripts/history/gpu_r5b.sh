#!/bin/bash
# round 4, call B: fused split-K reduce + LayerNorm (fuse_ln_reduce), 4-rows-per-wave LayerNorm (ln_r4), new flash defaults: tests + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "split_linear_reduce or layer_norm or linear or feed_forward or spatial or flash or attention" ) > gpurun_out/r5b_tests_default.log 2>&1; echo "rc=$?" >> gpurun_out/r5b_tests_default.log
( SDCPP_BACKEND_OPTS="ln_r4=1" timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "layer_norm or norm or unet or mmdit or flux or clip or t5 or spatial or feed" ) > gpurun_out/r5b_tests_ln_r4.log 2>&1; echo "rc=$?" >> gpurun_out/r5b_tests_ln_r4.log
( timeout 300 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "unet or sdxl" ) > gpurun_out/r5b_tests_models.log 2>&1; echo "rc=$?" >> gpurun_out/r5b_tests_models.log
for o in "fuse_ln_reduce=0" "fuse_ln_reduce=1" "fuse_ln_reduce=0 ln_r4=1" "fuse_ln_reduce=1 ln_r4=1"; do
  echo "#### sdxl $o" >> gpurun_out/r5b_family_sdxl.txt
  timeout 200 python scripts/family_times.py sdxl $o 2>&1 | head -12 >> gpurun_out/r5b_family_sdxl.txt
done
timeout 200 python scripts/ab_bench.py ln_r4 0,1 3 4 > gpurun_out/r5b_ab_ln_r4.txt 2>&1
timeout 200 python scripts/ab_bench.py fuse_ln_reduce 0,1 3 4 > gpurun_out/r5b_ab_fuse_ln_reduce.txt 2>&1
tail -n 4 gpurun_out/r5b_tests_*.log; grep "####\|==\|split-K\|LayerNorm" gpurun_out/r5b_family_sdxl.txt; tail -n 3 gpurun_out/r5b_ab_*.txt
