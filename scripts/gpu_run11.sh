run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
mkdir -p gpurun_out/r01c
T=300 TAILN=15 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=400 TAILN=2 run python bench.py --gpus 1 --steps 10 --warmup 2
export TMPDIR=/tmp
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r01c -o stats -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r01c/bench_under_rocprof.jsonl 2> $GRAFT_REPO_ROOT/gpurun_out/r01c/stderr.log )
tail -1 gpurun_out/r01c/bench_under_rocprof.jsonl | cut -c1-1500
python scripts/rocpd_stats.py gpurun_out/r01c/stats_results.db gpurun_out/r01c/kernel_stats.csv | head -8 | cut -c1-150
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_gemm16" -d $GRAFT_REPO_ROOT/gpurun_out/r01c -o pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r01c/pmc_fetch.log )
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "k_gemm16" -d $GRAFT_REPO_ROOT/gpurun_out/r01c -o pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r01c/pmc_write.log )
ls -la gpurun_out/r01c
python scripts/pmc_traffic.py gpurun_out/r01c/pmc_fetch_results.db gpurun_out/r01c/pmc_write_results.db "k_gemm16<128, true, 32, 3, 8>" gpurun_out/r01c/pmc_traffic.json
tail -3 gpurun_out/r01c/pmc_fetch.log
