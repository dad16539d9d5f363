#!/bin/bash
# round 3, call J: flash_vtr in the DiT / SDXL forwards: real-width block parity + per-family kernel time with and without
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 70 python -m pytest tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "real_width_flux_blocks_vs_oracle or real_width_sd35" ) > gpurun_out/r3J_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3J_tests.log
tail -3 gpurun_out/r3J_tests.log
for m in flux sd35; do for o in "flash_vtr=0" ""; do echo "== $m $o"; timeout 40 python scripts/family_times.py $m $o 2>&1 | head -4; done; done > gpurun_out/r3J_family_dit_vtr.txt 2>&1
cat gpurun_out/r3J_family_dit_vtr.txt
