# round 5, GPU call 5: multi-row LayerNorm -> f16 image kernel (op test, in-process A/B on the SD1.5 and SDXL forwards)
D=gpurun_out/r6e
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE
timeout 600 python -m pytest tests/test_gpu_ops.py -k "layer_norm or multi_row" -m gpu -x -q > $D/ln_tests.log 2>&1; echo "ln tests rc=$?"; tail -3 $D/ln_tests.log
timeout 600 python scripts/family_times.py sd15 ln16_rows=1 2>&1 | grep -E "==|LayerNorm" | tee $D/ln_rows1_sd15.txt
timeout 600 python scripts/family_times.py sd15 ln16_rows=4 2>&1 | grep -E "==|LayerNorm" | tee $D/ln_rows4_sd15.txt
timeout 600 python scripts/ab_bench.py ln16_rows 1,4 3 4 2>&1 | tail -3 | tee $D/ab_bench_ln16_rows.txt
timeout 600 python scripts/family_times.py sdxl ln16_rows=1 2>&1 | grep -E "==|LayerNorm" | tee $D/ln_rows1_sdxl.txt
timeout 600 python scripts/family_times.py sdxl ln16_rows=4 2>&1 | grep -E "==|LayerNorm" | tee $D/ln_rows4_sdxl.txt
