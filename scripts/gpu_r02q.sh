D=gpurun_out/r02q
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "flash or attention or unet" 2>&1 | grep -E "^E|assert|passed|failed" | head -20
for v in 0 1 0 1; do
MI355X_KTIME_DUMP=$D/shapes_mslot$v.txt timeout 500 python scripts/family_times.py sd15 flash_mslot=$v 2>&1 | grep -E "==|flash"
done
grep "343.597" $D/shapes_mslot*.txt
