# round 5, final GPU call: the whole -m gpu suite on the final code, the reference-faithful pixel leg once (printed), smoke, the default bench line, the same
# command under rocprofv3 --kernel-trace --stats, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the conv family and the bandwidth producers
D=gpurun_out/r6z
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; echo "bench rc=$?"; tail -c 600 $D/bench_default.jsonl
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$D -o stats -- python $R/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 > $R/$D/bench_under_rocprof.jsonl 2> $R/$D/stderr.log )
python scripts/rocpd_stats.py $D/stats_results.db $D/kernel_stats.csv | head -14 | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_conv3w|k_gemm16|k_nchw_to_nhwc|k_layer_norm_f16|k_gn_stats|k_fgemv" -d $R/$D -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --hip-graph 0 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 --no-e2e --no-kernels > /dev/null 2> $R/$D/pmc_$c.log )
done
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w|k_gemm16<256, %, true" $D/pmc_traffic_conv256.json | grep -E "hbm_bytes|launches"
python scripts/pmc_by_shape.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w|k_gemm16<256, %, true" $D/pmc_conv256_by_shape.txt | tail -3
for k in k_nchw_to_nhwc_f16 k_layer_norm_f16 k_gn_stats; do python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "$k" $D/pmc_traffic_$k.json | grep -E "hbm_bytes"; done
rm -f $D/*_results.db
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $D/suite.log 2>&1; echo "suite rc=$?"; tail -16 $D/suite.log
SDCPP_PIXELS_FAITHFUL=1 timeout 900 python -m pytest "tests/test_zz_gpu_pixels.py::test_sd15_20_step_euler_a_pixels_vs_oracle" -m gpu -q -s > $D/pixels_faithful.log 2>&1; echo "pixels rc=$?"; grep -E "rel-L2|PSNR|trajectory|faithful|passed|failed" $D/pixels_faithful.log | tail -12
