#!/bin/bash
# round 4, call Q: SDXL's 2048-token Linears: K slices + slab reduce against unsplit 128 x 64 tiles (bn64_max_tiles / splitk_target); GEGLU tile rule check on SD1.5 / SDXL
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -f gpurun_out/r5q_family_sdxl.txt
for rep in 1 2; do for o in "bn64_max_tiles=128 splitk_target=384" "bn64_max_tiles=192 splitk_target=256" "bn64_max_tiles=192 splitk_target=384" "bn64_max_tiles=128 splitk_target=256"; do echo "#### sdxl $o" >> gpurun_out/r5q_family_sdxl.txt; timeout 300 python scripts/family_times.py sdxl $o 2>&1 | head -7 >> gpurun_out/r5q_family_sdxl.txt; done; done
timeout 200 python scripts/family_times.py sd15 > gpurun_out/r5q_family_sd15.txt 2>&1
grep "####\|==\|Linear\|split-K" gpurun_out/r5q_family_sdxl.txt; head -4 gpurun_out/r5q_family_sd15.txt
