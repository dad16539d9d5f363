# round 5, GPU call 2: the tests call 1 did not reach (abi remap fixed; full-depth DiT parity incl. the faithful leg once; SDXL batch 8), oracle GEMM speed,
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts, the row-split / padded-tile policy (op test + A/B on FLUX and SD3.5), SDXL in-launch split-K A/B
D=gpurun_out/r6b
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -3
import time, numpy as np, sdcpp_amd as sd
sd.load_backend("oracle/_build/libggml-cpu-oracle.so")
e = sd.Engine(model=sd.SD15, backend="CPU-oracle")
x = np.random.default_rng(0).standard_normal((1, 4, 64, 64)).astype(np.float32); c = np.random.default_rng(1).standard_normal((1, 77, 768)).astype(np.float32)
e.unet_forward(x, np.array([500.0], np.float32), c)
t0 = time.perf_counter(); e.unet_forward(x, np.array([500.0], np.float32), c); print("oracle SD1.5 forward (round-4 loop: 2.73 s on this class of box):", round(time.perf_counter() - t0, 2), "s")
PY
timeout 600 python -m pytest tests/test_gpu_abi_remap.py "tests/test_gpu_ops.py::test_linear_row_split_and_padded_256_tiles" -m gpu -x -q -s > $D/new_tests_a.log 2>&1; echo "abi/tail tests rc=$?"; tail -4 $D/new_tests_a.log
# calibration of the HBM counters on known byte counts (1 GiB per launch, 3 timed launches + 1 warm per kernel)
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib scripts/fetch_calib.hip && /tmp/fetch_calib 1024 3 > $D/fetch_calib_rates.txt 2>&1; cat $D/fetch_calib_rates.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$D -o calib_$c -- /tmp/fetch_calib 1024 1 > /dev/null 2> $R/$D/calib_$c.log )
done
python scripts/fetch_calib.py $D/calib_FETCH_SIZE_results.db $D/calib_WRITE_SIZE_results.db $((1024*1024*1024)) $D/fetch_calib.json 2>&1 | tail -12
rm -f $D/*_results.db
# tile policy A/B (launch-time options, interleaved in one process)
timeout 900 python scripts/ab_family.py flux 3 -- tail_split=0,t256p_pad=0 -- tail_split=1,t256p_pad=1 2>&1 | tail -2 | tee $D/ab_flux_tail_split.txt
timeout 900 python scripts/ab_family.py sd35 3 -- tail_split=0,t256p_pad=0 -- tail_split=1,t256p_pad=0 -- tail_split=0,t256p_pad=1 -- tail_split=1,t256p_pad=1 2>&1 | tail -4 | tee $D/ab_sd35_tail_split.txt
timeout 600 python scripts/ab_family.py sdxl 3 -- tail_split=0,t256p_pad=0 -- tail_split=1,t256p_pad=1 2>&1 | tail -2 | tee $D/ab_sdxl_tail_split.txt
# the long oracle legs last
SDCPP_FULLDEPTH_FAITHFUL=1 timeout 2400 python -m pytest tests/test_zz_gpu_fulldepth.py "tests/test_zz_gpu_fullsize.py::test_sdxl_batch_8_on_one_gpu_vs_batch_1_oracle_trajectory" -m gpu -q -s --durations=5 > $D/fulldepth.log 2>&1; echo "fulldepth rc=$?"
grep -E "depth|growth|SDXL 1024|passed|failed|Error|assert" $D/fulldepth.log | tail -24
