D=gpurun_out/r02q
mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "flash or attention or unet or mmdit or flux" 2>&1 | tail -3
MI355X_KTIME_DUMP=$D/shapes_qvec.txt timeout 500 python scripts/family_times.py sd15 2>&1 | grep -E "==|flash|Linear"
grep flash $D/shapes_qvec.txt
timeout 500 python scripts/family_times.py sdxl 2>&1 | grep -E "==|flash|Linear"
