# A/B: flash attention (d <= 64 instantiations) compiled for 4 waves per SIMD against the default 3 (160 VGPRs).
# Build the variant library first (CPU box):  SDCPP_BUILD_VARIANT=occ4 python -c "import concurrent.futures as cf, sdcpp_amd.build as b; b.build_backend(cf.ThreadPoolExecutor(8))"
# Result of round 2 (profiles/r03e_flash_occupancy.txt): 128 VGPRs + ~25 registers spilled into the tile loop, 20-27 % slower.
R=$GRAFT_REPO_ROOT
O4=$R/stable-diffusion.cpp_amd/lib_occ4/libggml-mi355x.so
SDCPP_BACKEND_LIB=$O4 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "flash or attention" 2>&1 | tail -1
for rep in 1 2; do
for v in default occ4; do
  if [ $v = default ]; then unset SDCPP_BACKEND_LIB; else export SDCPP_BACKEND_LIB=$O4; fi
  echo "#### $v"
  timeout 300 python scripts/family_times.py sd15 2>&1 | grep -E "==|flash"
done
done
unset SDCPP_BACKEND_LIB
