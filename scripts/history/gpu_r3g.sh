#!/bin/bash
# round 3, call G: per-shape launch times of the bench forward (SD1.5) and the SDXL forward; flash ping-pong policy A/B on the bench forward
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( MI355X_KTIME_DUMP=gpurun_out/r3g_shapes_sd15.txt timeout 300 python scripts/family_times.py sd15 ) > gpurun_out/r3g_family_sd15.log 2>&1
( MI355X_KTIME_DUMP=gpurun_out/r3g_shapes_sdxl.txt timeout 300 python scripts/family_times.py sdxl ) > gpurun_out/r3g_family_sdxl.log 2>&1
( timeout 300 python scripts/ab_bench.py flash_pp 0,1,2 3 4 ) > gpurun_out/r3g_ab_flash_pp.log 2>&1
head -30 gpurun_out/r3g_family_sd15.log; tail -5 gpurun_out/r3g_ab_flash_pp.log
