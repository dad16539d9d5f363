D=gpurun_out/r02k
mkdir -p $D
for o in "" "fuse_chan_add=0" "fuse_proj_tokens=0" "fuse_chan_add=0,fuse_proj_tokens=0"; do
  echo "== opts: $o"
  SDCPP_BACKEND_OPTS=$o timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "tokens_to_image or vae_decode_parity" 2>&1 | grep -E "PSNR|passed|failed"
done
for rep in 1 2; do
for o in "fuse_chan_add=1" "fuse_chan_add=0" "fuse_proj_tokens=0" ; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sdxl --no-e2e --no-kernels --backend-opt $o > $D/bench_$o.jsonl 2>/dev/null; python -c "
import json; d=json.loads(open('$D/bench_$o.jsonl').read().strip().splitlines()[-1]); print('$o', d['ms_per_step'], d['host_loop']['ms_per_step'])"
done
done
