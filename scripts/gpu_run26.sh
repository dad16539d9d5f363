run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=400 TAILN=14 run python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "flux"
T=1200 TAILN=3 run python bench.py --model flux --steps 2 --warmup 1 --batch 1 --no-cpu-baseline
