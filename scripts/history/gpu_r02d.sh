D=gpurun_out/r02d
mkdir -p $D
timeout 900 python scripts/t320_check.py 2>&1 | tee $D/t320_check.txt | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sdxl > $D/bench.jsonl 2>$D/bench.err; python -c "
import json; d=json.loads(open('$D/bench.jsonl').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved']); [print(k['name'][:50], k['ms_per_step'], k['achieved']) for k in d['roofline']['kernels']]"
