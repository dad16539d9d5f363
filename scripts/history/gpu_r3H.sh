#!/bin/bash
# round 3, call H: flash attention with row-major V tiles + ds_read_b64_tr_b16 (flash_vtr): correctness and timing against the default kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 240 python scripts/flash_check.py vtr > gpurun_out/r3H_flash_vtr.txt 2>&1
echo "rc=$?" >> gpurun_out/r3H_flash_vtr.txt
cat gpurun_out/r3H_flash_vtr.txt
