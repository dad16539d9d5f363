#!/bin/bash
# round 3, call B: first run of the two-query-block flash kernel and the LDS-window conv kernel: check scripts (correctness + timing A/B), the op / model
# parity suites, a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 400 python scripts/flash_check.py ) > gpurun_out/r3b_flash_check.log 2>&1
( time timeout 500 python scripts/conv3w_check.py ) > gpurun_out/r3b_conv3w_check.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x \
    -k "not vs_oracle and not vae_mid" ) > gpurun_out/r3b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b_tests.log
( time timeout 300 python bench.py --skip-legs flux,sd35 --no-cpu-baseline ) > gpurun_out/r3b_bench.log 2>&1
tail -3 gpurun_out/r3b_flash_check.log; tail -3 gpurun_out/r3b_conv3w_check.log; tail -4 gpurun_out/r3b_tests.log; tail -c 600 gpurun_out/r3b_bench.log
