run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=300 TAILN=5 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=200 TAILN=1 run python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline
mkdir -p gpurun_out/prof10
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof10 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof10/bench.log 2> /dev/null )
python scripts/rocpd_stats.py gpurun_out/prof10/r_results.db gpurun_out/prof10/stats.csv | head -24 | cut -c1-150
