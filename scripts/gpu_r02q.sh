D=gpurun_out/r02q
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "flash or attention or unet or mmdit or flux or head_major" 2>&1 | grep -E "^E|assert|passed|failed" | head -20
for v in 0 1 0 1; do
timeout 500 python scripts/family_times.py sd15 fuse_q16=$v 2>&1 | grep -E "==|flash|Linear"
done
