mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
run() { echo "=== $*"; timeout $T "$@" 2>&1 | tail -12; echo "rc=$?"; }
T=400 run python -m pytest tests/test_gpu_model.py -m gpu -x -q -s
T=300 run python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline
T=300 run python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-flash
T=300 run python bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline
mkdir -p gpurun_out/prof1
( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1/bench.log 2>&1 )
ls -R gpurun_out/prof1 | head -20
