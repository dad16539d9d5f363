#!/bin/bash
# round 3, call P: FLUX q / k / v operand assembly (plan_flux_qkv): DiT tests + FLUX / SD3.5 family times on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "flux or Flux or dit or mmdit or sd3 or concat or joint or rope" ) > gpurun_out/r3p_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3p_tests.log
( timeout 600 python scripts/family_times.py flux ) > gpurun_out/r3p_family_flux.txt 2>&1
( timeout 600 python scripts/family_times.py flux fuse_joint_qkv=0 ) > gpurun_out/r3p_family_flux_off.txt 2>&1
tail -5 gpurun_out/r3p_tests.log; head -18 gpurun_out/r3p_family_flux.txt; head -3 gpurun_out/r3p_family_flux_off.txt
