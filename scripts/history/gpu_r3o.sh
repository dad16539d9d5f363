#!/bin/bash
# round 3, call O: MMDiT joint-attention operand assembly (plan_joint_qkv): DiT tests + SD3.5 family times on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "flux or Flux or dit or mmdit or sd3 or concat or joint" ) > gpurun_out/r3o_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3o_tests.log
( timeout 600 python scripts/family_times.py sd35 ) > gpurun_out/r3o_family_sd35.txt 2>&1
( timeout 600 python scripts/family_times.py sd35 fuse_joint_qkv=0 ) > gpurun_out/r3o_family_sd35_off.txt 2>&1
tail -5 gpurun_out/r3o_tests.log; head -16 gpurun_out/r3o_family_sd35.txt; head -3 gpurun_out/r3o_family_sd35_off.txt
