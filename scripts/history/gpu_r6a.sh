# round 5, GPU call 1: everything written before the first GPU minute of the round — enum remap through a shifted-enum host, hipGraph replay as the
# default, SDXL VAE Conv2d scale, full-depth DiT parity, SDXL batch 8, then the whole suite and the default bench line (self-launch, hip graph, new legs)
D=gpurun_out/r6a
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE
nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_abi_remap.py tests/test_zz_gpu_fulldepth.py "tests/test_gpu_model.py::test_vae_decode_with_conv2d_scale_parity" \
   "tests/test_zz_gpu_fullsize.py::test_full_size_sdxl_vae_decode_with_conv2d_scale_vs_oracle" "tests/test_zz_gpu_fullsize.py::test_sdxl_batch_8_on_one_gpu_vs_batch_1_oracle_trajectory" \
   -m gpu -x -q -s --durations=8 > $D/new_tests.log 2>&1; echo "new tests rc=$?"; grep -E "depth|growth|PSNR|SDXL|passed|failed|Error|error|rc=" $D/new_tests.log | tail -30
SDCPP_SKIP_FULLDEPTH=1 timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 --deselect tests/test_gpu_abi_remap.py \
   --deselect "tests/test_zz_gpu_fullsize.py::test_sdxl_batch_8_on_one_gpu_vs_batch_1_oracle_trajectory" > $D/suite.log 2>&1; echo "suite rc=$?"; tail -25 $D/suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; echo "bench rc=$?"; tail -c 3000 $D/bench_default.jsonl; tail -5 $D/bench_default.err
