D=gpurun_out/r02h
mkdir -p $D
SDCPP_BACKEND_LIB=$PWD/stable-diffusion.cpp_amd/lib_exp/libggml-mi355x.so timeout 600 python scripts/gemm_ablation.py 2>&1 | tee $D/gemm_ablation.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
