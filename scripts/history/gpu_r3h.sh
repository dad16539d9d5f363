#!/bin/bash
# round 3, call H: 1x1-conv tile probe, FF2 -> f16 operand rows (test + A/B on the bench forward)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python scripts/conv1x1_probe.py ) > gpurun_out/r3h_conv1x1_probe.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "ff2_residual or spatial_transformer or linear" ) > gpurun_out/r3h_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3h_tests.log
( timeout 300 python scripts/ab_bench.py fuse_rows16 0,1 3 4 ) > gpurun_out/r3h_ab_rows16.log 2>&1
cat gpurun_out/r3h_conv1x1_probe.log | cut -c1-400; tail -4 gpurun_out/r3h_tests.log; tail -3 gpurun_out/r3h_ab_rows16.log
