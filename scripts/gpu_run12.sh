run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
mkdir -p gpurun_out/r01d
T=300 TAILN=5 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=400 TAILN=1 run python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_gemm16" -d $GRAFT_REPO_ROOT/gpurun_out/r01d -o pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r01d/pmc_fetch.log )
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "k_gemm16" -d $GRAFT_REPO_ROOT/gpurun_out/r01d -o pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r01d/pmc_write.log )
python scripts/pmc_traffic.py gpurun_out/r01d/pmc_fetch_results.db gpurun_out/r01d/pmc_write_results.db "k_gemm16<128, true, 32, 3, 8>" gpurun_out/r01d/pmc_traffic.json | head -12
