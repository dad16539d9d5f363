mkdir -p gpurun_out/prof6
( cd /tmp && export TMPDIR=/tmp && GGML_MI355X_TRACE=1 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof6 -o r6 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof6/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/prof6/trace.log )
grep -c "^G16" gpurun_out/prof6/trace.log
python scripts/shape_stats.py gpurun_out/prof6/r6_results.db gpurun_out/prof6/trace.log | cut -c1-170
