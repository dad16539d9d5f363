mkdir -p gpurun_out/flux
( cd /tmp && export TMPDIR=/tmp && GGML_MI355X_TRACE=1 timeout 1200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/flux -o r -- python $GRAFT_REPO_ROOT/bench.py --model flux --steps 1 --warmup 1 --batch 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/flux/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/flux/trace.log )
python scripts/rocpd_stats.py gpurun_out/flux/r_results.db gpurun_out/flux/stats.csv | head -24 | cut -c1-150
