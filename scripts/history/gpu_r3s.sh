#!/bin/bash
# round 3, call S: pipelined 128x128 tile (T128P): op tests, SDXL / SD1.5 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x ) > gpurun_out/r3s_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3s_tests.log
for o in "gemm16_t128p=0" ""; do echo "== opts: $o"; timeout 300 python scripts/family_times.py sdxl $o 2>&1 | head -8; done > gpurun_out/r3s_sdxl_t128p.txt 2>&1
( timeout 300 python scripts/ab_bench.py gemm16_t128p 0,1 3 4 ) > gpurun_out/r3s_ab_t128p.log 2>&1
tail -4 gpurun_out/r3s_tests.log; cat gpurun_out/r3s_sdxl_t128p.txt; tail -3 gpurun_out/r3s_ab_t128p.log
