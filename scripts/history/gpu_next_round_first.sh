#!/bin/bash
# First GPU call of the next round: the variants that were written after this round's GPU budget was spent (all option-gated OFF):
#   gemm16_swp  = 1  Linear kernels of the 256-row tiles with the MFMA operands swapped: transposed accumulator, 16-byte epilogue accesses
#   flash_short = 1  k_flash_short: K / V register-resident kernel for Lk <= 96 (the 77-token cross-attention), d <= 64
#   flash_nsel  = 1  select-free K / V staging (d = 40 two-block, d = 64, d = 128 kernels)
#   flash_ovl   = 2  overlapped issue order for the d <= 48 launches without the max slot (with flash_qb2 = 2)
# Correctness (sampled rows against float64 softmax, variants against each other) + HIP-event timing, alternating with the default kernels;
# then the attention tests and the model tests with each option forced on through SDCPP_BACKEND_OPTS.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 120 python scripts/flash_check.py short "Lk" > gpurun_out/next_flash_short.txt 2>&1; echo "rc=$?" >> gpurun_out/next_flash_short.txt
timeout 120 python scripts/flash_check.py nsel > gpurun_out/next_flash_nsel.txt 2>&1; echo "rc=$?" >> gpurun_out/next_flash_nsel.txt
timeout 120 python scripts/flash_check.py ovl2 "d" > gpurun_out/next_flash_ovl2.txt 2>&1; echo "rc=$?" >> gpurun_out/next_flash_ovl2.txt
for o in "flash_short=1" "flash_nsel=1"; do
  ( SDCPP_BACKEND_OPTS="$o" timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x -k "flash or attention or attn or unet" ) > "gpurun_out/next_tests_${o%%=*}.log" 2>&1
  echo "tests rc=$?" >> "gpurun_out/next_tests_${o%%=*}.log"
done
# gemm16_swp = 1: big-token Linear tiles with the accumulator transposed (16-byte epilogue accesses): every Linear / GEGLU / model test with it on, then A/B
( SDCPP_BACKEND_OPTS="gemm16_swp=1" timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x -k "linear or geglu or feed or ff or unet or mmdit or flux" ) > gpurun_out/next_tests_gemm16_swp.log 2>&1
echo "tests rc=$?" >> gpurun_out/next_tests_gemm16_swp.log
for o in flash_short flash_nsel gemm16_swp; do timeout 200 python scripts/ab_bench.py $o 0,1 3 4 > gpurun_out/next_ab_$o.txt 2>&1; done
tail -n 14 gpurun_out/next_flash_short.txt gpurun_out/next_flash_nsel.txt gpurun_out/next_flash_ovl2.txt; tail -n 3 gpurun_out/next_tests_*.log gpurun_out/next_ab_*.txt
