# round 5, GPU call 3: the whole GPU suite with the AVX-512 oracle (wall time of the suite is oracle time), where the conv family's 2.1x counter traffic
# comes from (per-shape PMC join, launch trace), the default bench line under rocprofv3 --stats
D=gpurun_out/r6c
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=14 > $D/suite.log 2>&1; echo "suite rc=$?"; tail -24 $D/suite.log
# conv family by launch shape: counters (separate passes) + the launch trace of the same command
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_conv3w|k_gemm16" -d $R/$D -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --hip-graph 0 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 --no-e2e --no-kernels > /dev/null 2> $R/$D/pmc_$c.log )
done
python scripts/pmc_by_shape.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w|k_gemm16<256, %, true" $D/pmc_conv256_by_shape.txt | tail -45
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_conv3w|k_gemm16<256, %, true" $D/pmc_traffic_conv256.json | grep -E "hbm_bytes|launches"
python scripts/pmc_by_shape.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_gemm16<%false" $D/pmc_linear_by_shape.txt | tail -30
GGML_MI355X_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 0 --hip-graph 0 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 --no-e2e --no-kernels 2>&1 | grep -E "^G16 conv|^conv3w|^G16 linear" | sort | uniq -c | sort -rn > $D/launch_trace_sd15.txt; head -40 $D/launch_trace_sd15.txt
rm -f $D/*_results.db
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; echo "bench rc=$?"; tail -c 1500 $D/bench_default.jsonl
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$D -o stats -- python $R/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 > $R/$D/bench_under_rocprof.jsonl 2> $R/$D/stderr.log )
python scripts/rocpd_stats.py $D/stats_results.db $D/kernel_stats.csv | head -16 | cut -c1-170
rm -f $D/*_results.db
