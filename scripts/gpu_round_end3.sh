# end-of-round measurement sequence without the test suite: smoke, default bench line, rocprofv3 kernel summary of the same command,
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the dominant conv kernel and the bandwidth-bound producers
D=${D:-gpurun_out/r04}
mkdir -p $D
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; tail -c 600 $D/bench_default.jsonl
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $R/$D -o stats -- python $R/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 > $R/$D/bench_under_rocprof.jsonl 2> $R/$D/stderr.log )
python scripts/rocpd_stats.py $D/stats_results.db $D/kernel_stats.csv | head -14 | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_conv3w|k_gemm16|k_nchw_to_nhwc|k_layer_norm_f16|k_gn_stats|k_fgemv" -d $R/$D -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 --no-e2e --no-kernels > /dev/null 2> $R/$D/pmc_$c.log )
done
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w|k_gemm16<256, %, true" $D/pmc_traffic_conv256.json | grep -E "hbm_bytes|launches"
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w" $D/pmc_traffic_conv3w.json | grep -E "hbm_bytes|launches"
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_nchw_to_nhwc_f16" $D/pmc_traffic_nchw_to_nhwc.json | grep -E "hbm_bytes|launches"
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_layer_norm_f16" $D/pmc_traffic_layer_norm_f16.json | grep -E "hbm_bytes|launches"
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_gn_stats" $D/pmc_traffic_gn_stats.json | grep -E "hbm_bytes|launches"
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_fgemv" $D/pmc_traffic_fgemv.json | grep -E "hbm_bytes|launches"
rm -f $D/*_results.db
