run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=300 TAILN=5 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=400 TAILN=5 run python scripts/ab_bench.py gemm16_tile -1,3,4,1 2 3
