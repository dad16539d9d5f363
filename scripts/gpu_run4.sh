run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=200 TAILN=4 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=400 TAILN=60 run python scripts/microbench.py 0 1 2
for v in 0 1 2; do T=200 TAILN=1 run python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --g16-variant $v; done
T=200 TAILN=1 run python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --hip-graph 1
