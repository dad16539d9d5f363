run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=300 TAILN=15 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=200 TAILN=1 run python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline
mkdir -p gpurun_out/prof9
( cd /tmp && export TMPDIR=/tmp && GGML_MI355X_TRACE=1 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof9 -o r9 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof9/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/prof9/trace.log )
python scripts/shape_stats.py gpurun_out/prof9/r9_results.db gpurun_out/prof9/trace.log | cut -c1-170 | head -40
python scripts/rocpd_stats.py gpurun_out/prof9/r9_results.db gpurun_out/prof9/stats.csv | head -16 | cut -c1-150
