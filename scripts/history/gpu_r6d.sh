# round 5, GPU call 4: weight-major workgroup order of the weight-heavy convs (8x8 / 16x16 UNet levels): op test, in-process A/B on the SD1.5 forward, counter
# traffic per launch shape with it on; where an oracle forward spends its time
D=gpurun_out/r6d
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest "tests/test_gpu_ops.py::test_conv2d_weight_major_workgroup_order" tests/test_gpu_ops.py -k "conv" -m gpu -x -q > $D/conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -3 $D/conv_tests.log
timeout 600 python scripts/ab_family.py sd15 5 -- conv_wmajor=0 -- conv_wmajor=1 2>&1 | tail -2 | tee $D/ab_sd15_conv_wmajor.txt
timeout 600 python scripts/ab_bench.py conv_wmajor 0,1 3 4 2>&1 | tail -6 | tee $D/ab_bench_conv_wmajor.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_conv3w|k_gemm16" -d $R/$D -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --hip-graph 0 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 --no-e2e --no-kernels > /dev/null 2> $R/$D/pmc_$c.log )
done
python scripts/pmc_by_shape.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w|k_gemm16<256, %, true" $D/pmc_conv256_by_shape.txt | tail -16
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w|k_gemm16<256, %, true" $D/pmc_traffic_conv256.json | grep -E "hbm_bytes|launches"
rm -f $D/*_results.db
ORACLE_PROFILE=1 python - <<'PY' 2>&1 | tail -3
import time, numpy as np, sdcpp_amd as sd
sd.load_backend("oracle/_build/libggml-cpu-oracle.so")
e = sd.Engine(model=sd.SD15, backend="CPU-oracle")
x = np.random.default_rng(0).standard_normal((1, 4, 64, 64)).astype(np.float32); c = np.random.default_rng(1).standard_normal((1, 77, 768)).astype(np.float32)
e.unet_forward(x, np.array([500.0], np.float32), c)
t0 = time.perf_counter(); e.unet_forward(x, np.array([500.0], np.float32), c); print("oracle SD1.5 forward:", round(time.perf_counter() - t0, 2), "s")
PY
