export TMPDIR=/tmp
for v in 4 -1; do
mkdir -p gpurun_out/tile$v
cat > /tmp/run_tile.py <<PYEOF
import sys
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"]
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import sdcpp_amd as sd
sd.backend_set_option("gemm16_tile", $v)
import runpy
runpy.run_path("$GRAFT_REPO_ROOT/bench.py", run_name="__main__")
PYEOF
( cd /tmp && GGML_MI355X_TRACE=1 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tile$v -o r -- python /tmp/run_tile.py > $GRAFT_REPO_ROOT/gpurun_out/tile$v/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/tile$v/trace.log )
python scripts/shape_stats.py gpurun_out/tile$v/r_results.db gpurun_out/tile$v/trace.log > gpurun_out/tile$v/shapes.txt
tail -1 gpurun_out/tile$v/shapes.txt
done
