#!/bin/bash
# round 4, call D: splitk_inkernel = 2 (in-launch split-K for head-major / f16 outputs only) on SDXL / SD1.5; then the whole GPU suite with the new defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -f gpurun_out/r5d_family_sdxl.txt
for o in "splitk_inkernel=0" "splitk_inkernel=2" "splitk_inkernel=0" "splitk_inkernel=2"; do
  echo "#### sdxl $o" >> gpurun_out/r5d_family_sdxl.txt
  timeout 200 python scripts/family_times.py sdxl $o 2>&1 | head -8 >> gpurun_out/r5d_family_sdxl.txt
done
timeout 200 python scripts/ab_bench.py splitk_inkernel 0,2 3 4 > gpurun_out/r5d_ab_splitk_inkernel.txt 2>&1
( SDCPP_BACKEND_OPTS="splitk_inkernel=2" timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "linear or head_major or projection or unet or attention" ) > gpurun_out/r5d_tests_inkernel2.log 2>&1; echo "rc=$?" >> gpurun_out/r5d_tests_inkernel2.log
( timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/r5d_tests_all.log 2>&1; echo "rc=$?" >> gpurun_out/r5d_tests_all.log
grep "####\|==\|Linear" gpurun_out/r5d_family_sdxl.txt; tail -n 3 gpurun_out/r5d_ab_splitk_inkernel.txt; tail -n 4 gpurun_out/r5d_tests_inkernel2.log gpurun_out/r5d_tests_all.log
