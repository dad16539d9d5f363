run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=400 TAILN=4 run python -m pytest tests -m gpu -x -q
T=900 TAILN=1 run python bench.py --model flux --steps 2 --warmup 1 --batch 1 --no-cpu-baseline
