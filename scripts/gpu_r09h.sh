# round 6, call r09h: encoder / img2img GPU tests (tiny + full size), encoder through the reference runner on the GPU
D=gpurun_out/r09h; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_ref_graphs.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -s -k "vae_encode or VAE_ENC" > $D/new_tests.log 2>&1; tail -4 $D/new_tests.log
grep -E "VAE encode|VAE_ENC.*rel-L2" $D/new_tests.log | cut -c1-260
