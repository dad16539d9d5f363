# per-shape k_gemm16 times of the SD1.5 batch-8 bench step (eager launches, GGML_MI355X_TRACE joined with a rocprofv3 kernel trace)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; D=gpurun_out/${1:-r09d}; mkdir -p $D
( cd /tmp && GGML_MI355X_TRACE=1 timeout 400 rocprofv3 --kernel-trace -d $R/$D -o tr -- python $R/bench.py --steps 2 --warmup 1 --hip-graph 0 --no-cpu-baseline --no-e2e --no-sdxl --skip-legs sdxl,flux,sd35,sdxl_b8 --no-kernels ${2:-} > $R/$D/tr.out 2> $R/$D/tr.err ); tail -1 $D/tr.out | cut -c1-300
python scripts/shape_stats.py $D/tr_results.db $D/tr.err > $D/shape_stats.txt 2>&1; head -60 $D/shape_stats.txt
rm -f $D/tr_results.db; grep "^G16" $D/tr.err | sort | uniq -c | sort -rn | head -5; rm -f $D/tr.err
