run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=400 TAILN=4 run python -m pytest tests -m gpu -x -q
T=300 TAILN=1 run python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline
