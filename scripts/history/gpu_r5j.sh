#!/bin/bash
# round 4, call J: relax_res_overlap (Linear + residual fusion when the sum sits on the Linear input's recycled buffer): unfused node list, model tests, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
GGML_MI355X_PLAN_TRACE=1 timeout 200 python scripts/family_times.py sd15 2> gpurun_out/r5j_plan_trace_sd15.txt > gpurun_out/r5j_family_sd15.txt
( timeout 500 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py tests/test_gpu_ops.py -m gpu -q -x -k "unet or sdxl or linear or transformer or attention or feed or mmdit or flux or sampler" ) > gpurun_out/r5j_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5j_tests.log
timeout 200 python scripts/ab_bench.py relax_res_overlap 0,1 3 4 > gpurun_out/r5j_ab_relax.txt 2>&1
for o in "relax_res_overlap=0" "relax_res_overlap=1"; do echo "#### sdxl $o"; timeout 200 python scripts/family_times.py sdxl $o 2>&1 | head -8; done > gpurun_out/r5j_family_sdxl.txt
grep -c "op 2 " gpurun_out/r5j_plan_trace_sd15.txt; head -12 gpurun_out/r5j_family_sd15.txt; tail -n 4 gpurun_out/r5j_tests.log gpurun_out/r5j_ab_relax.txt; grep "####\|==\|Linear\|binary" gpurun_out/r5j_family_sdxl.txt
