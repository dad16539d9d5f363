export TMPDIR=/tmp
for v in 1 3; do
mkdir -p gpurun_out/var$v
( cd /tmp && GGML_MI355X_TRACE=1 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/var$v -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --g16-variant $v > $GRAFT_REPO_ROOT/gpurun_out/var$v/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/var$v/trace.log )
python scripts/shape_stats.py gpurun_out/var$v/r_results.db gpurun_out/var$v/trace.log > gpurun_out/var$v/shapes.txt
tail -1 gpurun_out/var$v/shapes.txt
done
