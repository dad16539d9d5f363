run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=$?"; }
mkdir -p gpurun_out/flux2
T=600 TAILN=1 run python bench.py --model sdxl --steps 2 --warmup 1 --batch 4 --no-cpu-baseline
( cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/flux2 -o r -- python $GRAFT_REPO_ROOT/bench.py --model flux --steps 2 --warmup 1 --batch 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/flux2/bench.log 2> /dev/null )
tail -1 gpurun_out/flux2/bench.log | cut -c1-400
python scripts/rocpd_stats.py gpurun_out/flux2/r_results.db gpurun_out/flux2/stats.csv | head -22 | cut -c1-150
