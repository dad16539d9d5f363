#!/bin/bash
# round 3, call G: split-K slab reduce that also writes the next GroupNorm's statistics (fuse_gn_stats): targeted parity tests (+ the UNet / VAE model tests)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "split_conv_reduce or group_norm or conv2d_split or time_embedding or unet or vae" ) > gpurun_out/r3G2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3G2_tests.log
( timeout 300 python -m pytest tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "sd15 or unet or pixel" ) > gpurun_out/r3G2_fullsize.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3G2_fullsize.log
tail -4 gpurun_out/r3G2_tests.log; tail -4 gpurun_out/r3G2_fullsize.log
