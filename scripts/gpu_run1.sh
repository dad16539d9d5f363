set -x
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-flash 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline 2>&1 | tail -3
mkdir -p gpurun_out/prof1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1/bench.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head -20
