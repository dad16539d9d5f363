#!/bin/bash
# round 3, call A: new parity tests at the benchmarked shapes + the 20-step pixel test + bench with the configs 3-5 sub-records
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 1300 python -m pytest tests/test_zz_gpu_config_shapes.py tests/test_zz_gpu_pixels.py \
    "tests/test_zz_gpu_fullsize.py::test_full_width_sdxl_unet_q8_0_vs_oracle" \
    "tests/test_zz_gpu_fullsize.py::test_real_width_sd35_joint_blocks_vs_oracle" \
    "tests/test_zz_gpu_fullsize.py::test_real_width_flux_blocks_vs_oracle" \
    "tests/test_zz_gpu_fullsize.py::test_full_size_vae_decode_vs_oracle" \
    -m gpu -q -s --durations=25 ) > gpurun_out/r3a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a_tests.log
( time timeout 600 python bench.py ) > gpurun_out/r3a_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r3a_bench.log
tail -5 gpurun_out/r3a_tests.log
tail -c 1500 gpurun_out/r3a_bench.log
