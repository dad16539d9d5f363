# round 5, GPU call after the final set: the two planner edits made after it (backstop for elided SCALE / UPSCALE, direct-consumer rule of the Conv2d-scale look-through)
D=gpurun_out/r6f
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_abi_remap.py "tests/test_zz_gpu_fullsize.py::test_full_size_sdxl_vae_decode_with_conv2d_scale_vs_oracle" "tests/test_zz_gpu_fullsize.py::test_full_size_vae_decode_vs_oracle" -m gpu -x -q > $D/tests.log 2>&1; echo "rc=$?"; tail -4 $D/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-legs flux,sd35,sdxl_b8 2>/dev/null | python -c "
import sys, json
p = json.loads(sys.stdin.readline()); print(p['value'], p['ms_per_step'], p['sdxl']['ms_per_step'], p['sdxl']['vae_decode_ms'], p['sdxl']['sec_per_image'])"
