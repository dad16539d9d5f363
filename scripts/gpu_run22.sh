run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=300 TAILN=6 run python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "mmdit"
T=300 TAILN=8 run python -m pytest tests/test_gpu_model.py -m gpu -q -s
