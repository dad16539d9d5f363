run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
T=400 TAILN=12 run python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "flux"
T=400 TAILN=4 run python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q
T=1200 TAILN=1 run python bench.py --model flux --steps 2 --warmup 1 --batch 1 --no-cpu-baseline
mkdir -p gpurun_out/flux
( cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/flux -o r -- python $GRAFT_REPO_ROOT/bench.py --model flux --steps 1 --warmup 1 --batch 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/flux/bench.log 2> /dev/null )
python scripts/rocpd_stats.py gpurun_out/flux/r_results.db gpurun_out/flux/stats.csv | head -22 | cut -c1-150
