mkdir -p gpurun_out
run() { echo "=== $*"; timeout 150 "$@" 2>&1 | tail -8; echo "rc=$?"; }
run python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "layout or geglu"
run python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "flash_attn_ext"
run python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "manual_attention"
run python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "unet_forward_parity"
run python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "vae or sampler"
