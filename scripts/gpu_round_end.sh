run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
D=gpurun_out/r01h
mkdir -p $D
T=400 TAILN=4 run python -m pytest tests -m gpu -x -q
T=200 TAILN=2 run python -c "import __graft_entry__ as g; g.smoke()"
T=400 TAILN=1 run python bench.py --gpus 1 --steps 10 --warmup 2
python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $D/bench_default_nocpu.jsonl 2>/dev/null
export TMPDIR=/tmp
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$D -o stats -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$D/bench_under_rocprof.jsonl 2> $GRAFT_REPO_ROOT/$D/stderr.log )
tail -1 $D/bench_under_rocprof.jsonl | cut -c1-1200
python scripts/rocpd_stats.py $D/stats_results.db $D/kernel_stats.csv | head -12 | cut -c1-150
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_gemm16" -d $GRAFT_REPO_ROOT/$D -o pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$D/pmc_fetch.log )
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "k_gemm16" -d $GRAFT_REPO_ROOT/$D -o pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$D/pmc_write.log )
python scripts/pmc_traffic.py $D/pmc_fetch_results.db $D/pmc_write_results.db "k_gemm16<256, 160, true" $D/pmc_traffic_t160.json | head -12
python scripts/pmc_traffic.py $D/pmc_fetch_results.db $D/pmc_write_results.db "k_gemm16<256, 1" $D/pmc_traffic_256rows.json | grep hbm_bytes
