run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=400 TAILN=14 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=300 TAILN=1 run python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline
