D=gpurun_out/r02q
mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
timeout 500 python scripts/family_times.py sd15 gemm16_bn64=$v 2>&1 | grep -E "==|Linear"
done
for v in 0 1; do
timeout 500 python scripts/family_times.py sdxl gemm16_bn64=$v 2>&1 | grep -E "==|Linear"
done
