D=gpurun_out/r02f
mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "quantised_gemv or linear_weight" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_zz_gpu_fullsize.py -m gpu -q -s -k "flux or sd35" 2>&1 | grep -E "rel-L2|passed|failed|rror" | tail
SDCPP_INIT_TIMING=1 timeout 900 python bench.py --model flux --batch 1 --steps 4 --warmup 1 --no-e2e --no-cpu-baseline > $D/bench_flux.jsonl 2> $D/bench_flux.err; tail -c 3000 $D/bench_flux.jsonl; tail -5 $D/bench_flux.err
python - <<'PY'
import sys; sys.path.insert(0,'.')
import sdcpp_amd as sd
print({k:v for k,v in sd.backend_stats().items()})
PY
