run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=400 TAILN=4 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=900 TAILN=1 run python bench.py --model sd35 --steps 2 --warmup 1 --batch 1 --no-cpu-baseline
mkdir -p gpurun_out/sd35
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/sd35 -o r -- python $GRAFT_REPO_ROOT/bench.py --model sd35 --steps 1 --warmup 1 --batch 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/sd35/bench.log 2> /dev/null )
python scripts/rocpd_stats.py gpurun_out/sd35/r_results.db gpurun_out/sd35/stats.csv | head -16 | cut -c1-150
