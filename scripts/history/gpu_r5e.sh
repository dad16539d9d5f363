#!/bin/bash
# round 4, call E (first call of the second session): round-end measurement set of the current defaults (bench line, rocprofv3 kernel summary,
# PMC traffic) into gpurun_out/r05a, per-shape kernel times of the SD1.5 forward, then the never-run gemm16_swp variant: tests + step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
D=gpurun_out/r05a bash scripts/gpu_round_end3.sh > gpurun_out/r5e_round_end.log 2>&1
rm -f gpurun_out/r5e_shapes_sd15.txt
MI355X_KTIME_DUMP=gpurun_out/r5e_shapes_sd15.txt timeout 200 python scripts/family_times.py sd15 > gpurun_out/r5e_family_sd15.txt 2>&1
( SDCPP_BACKEND_OPTS="gemm16_swp=1" timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x -k "linear or geglu or feed or ff or unet or mmdit or flux" ) > gpurun_out/r5e_tests_gemm16_swp.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5e_tests_gemm16_swp.log
timeout 200 python scripts/ab_bench.py gemm16_swp 0,1 3 4 > gpurun_out/r5e_ab_gemm16_swp.txt 2>&1
tail -n 30 gpurun_out/r5e_round_end.log | cut -c1-300; tail -n 4 gpurun_out/r5e_tests_gemm16_swp.log gpurun_out/r5e_ab_gemm16_swp.txt; head -24 gpurun_out/r5e_family_sd15.txt
