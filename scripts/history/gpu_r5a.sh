#!/bin/bash
# round 4, call A: k_flash_short after the store fix (+ Q prefetch variant), per-shape times of the current SD1.5 / SDXL forwards
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 120 python scripts/flash_short_debug.py > gpurun_out/r5a_short_debug.txt 2>&1; echo "rc=$?" >> gpurun_out/r5a_short_debug.txt
timeout 150 python scripts/flash_check.py short "Lk" > gpurun_out/r5a_flash_short.txt 2>&1; echo "rc=$?" >> gpurun_out/r5a_flash_short.txt
for o in "flash_short=1" "flash_short=2"; do
  ( SDCPP_BACKEND_OPTS="$o" timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x -k "flash or attention or attn or unet" ) > "gpurun_out/r5a_tests_${o/=/_}.log" 2>&1
  echo "tests rc=$?" >> "gpurun_out/r5a_tests_${o/=/_}.log"
done
timeout 200 python scripts/ab_bench.py flash_short 0,1,2 3 4 > gpurun_out/r5a_ab_flash_short.txt 2>&1
rm -f gpurun_out/r5a_shapes_sd15.txt gpurun_out/r5a_shapes_sdxl.txt
MI355X_KTIME_DUMP=gpurun_out/r5a_shapes_sd15.txt timeout 200 python scripts/family_times.py sd15 flash_nsel=1 > gpurun_out/r5a_family_sd15.txt 2>&1
MI355X_KTIME_DUMP=gpurun_out/r5a_shapes_sdxl.txt timeout 200 python scripts/family_times.py sdxl flash_nsel=1 > gpurun_out/r5a_family_sdxl.txt 2>&1
tail -n 12 gpurun_out/r5a_short_debug.txt gpurun_out/r5a_flash_short.txt; tail -n 4 gpurun_out/r5a_tests_*.log gpurun_out/r5a_ab_flash_short.txt; head -30 gpurun_out/r5a_family_sd15.txt
