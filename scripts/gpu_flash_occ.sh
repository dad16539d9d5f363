# A/B: flash attention (d <= 64 instantiations) build variants against the default (3 waves per SIMD, 160 VGPRs, 64-key softmax):
#   occ3h = two-half online softmax (FA_HALF) at 3 waves per SIMD, occ4h = the same compiled for 4 waves per SIMD
R=$GRAFT_REPO_ROOT
for v in occ3h occ4h; do
SDCPP_BACKEND_LIB=$R/stable-diffusion.cpp_amd/lib_$v/libggml-mi355x.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "flash or attention" 2>&1 | tail -1
done
for rep in 1 2; do
for v in default occ3h occ4h; do
  if [ $v = default ]; then unset SDCPP_BACKEND_LIB; else export SDCPP_BACKEND_LIB=$R/stable-diffusion.cpp_amd/lib_$v/libggml-mi355x.so; fi
  echo "#### $v"
  timeout 300 python scripts/family_times.py sd15 2>&1 | grep -E "==|flash"
done
done
unset SDCPP_BACKEND_LIB
