#!/bin/bash
# round 3, call F2: just-in-time quantised weight images (jit_qimages): test + FLUX / SDXL in the resident-quantised mode
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "just_in_time or quantised" ) > gpurun_out/r3F_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3F_tests.log
for o in "" "jit_qimages=1" "jit_qimages=1 qgemm16_max_rows=0"; do echo "== flux $o"; timeout 400 python scripts/family_times.py flux $o 2>&1 | head -7; done > gpurun_out/r3F_flux_jit.txt 2>&1
for o in "" "jit_qimages=1"; do echo "== sdxl $o"; timeout 400 python scripts/family_times.py sdxl $o 2>&1 | head -4; done >> gpurun_out/r3F_flux_jit.txt 2>&1
tail -4 gpurun_out/r3F_tests.log; cat gpurun_out/r3F_flux_jit.txt
