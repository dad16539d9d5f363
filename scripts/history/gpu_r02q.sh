D=gpurun_out/r02q
mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5
for v in 0 1; do
timeout 500 python scripts/family_times.py sd15 splitk_inkernel=$v 2>&1 | grep -E "==|Linear|split|conv"
done
for v in 0 1; do
timeout 500 python scripts/family_times.py sdxl splitk_inkernel=$v 2>&1 | grep -E "==|Linear|split|conv"
done
