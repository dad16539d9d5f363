D=gpurun_out/r02n
mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "manual_attention or conv2d" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -x -q -s -k "vae" 2>&1 | grep -E "PSNR|passed|failed"
timeout 600 python scripts/family_times.py vae 2>&1 | tee $D/family_times_vae.txt
