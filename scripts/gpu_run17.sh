run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
mkdir -p gpurun_out/r01f
T=400 TAILN=1 run python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --e2e
T=600 TAILN=1 run python bench.py --model sdxl --steps 2 --warmup 1 --batch 4 --no-cpu-baseline
export TMPDIR=/tmp
( cd /tmp && timeout 500 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r01f -o e2e -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 8 --no-cpu-baseline --e2e > $GRAFT_REPO_ROOT/gpurun_out/r01f/e2e.log 2>&1 )
python scripts/rocpd_stats.py gpurun_out/r01f/e2e_results.db gpurun_out/r01f/e2e_stats.csv | head -20 | cut -c1-150
