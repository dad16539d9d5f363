#!/bin/bash
# round 3, call I: flash_vtr on by default: attention parity tests (ops, benchmarked shapes, tiny + real-width models) and the SD1.5 step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_zz_gpu_config_shapes.py tests/test_gpu_model.py -m gpu -q -x -k "flash or attention or attn or unet or mmdit or flux or vae" ) > gpurun_out/r3I_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3I_tests.log
tail -3 gpurun_out/r3I_tests.log
timeout 200 python scripts/ab_bench.py flash_vtr 0,31 3 4 > gpurun_out/r3I_ab_flash_vtr.txt 2>&1
tail -3 gpurun_out/r3I_ab_flash_vtr.txt
