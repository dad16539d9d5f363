run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=200 TAILN=6 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=200 TAILN=1 run python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline
mkdir -p gpurun_out/prof5
( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -o r5 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof5/bench.log 2>&1 )
python scripts/rocpd_stats.py gpurun_out/prof5/r5_results.db gpurun_out/prof5/stats.csv | head -24 | cut -c1-160
