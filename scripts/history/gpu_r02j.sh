D=gpurun_out/r02j
mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sdxl > $D/bench.jsonl 2>$D/bench.err; python -c "
import json; d=json.loads(open('$D/bench.jsonl').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'], d['backend']); [print(k['name'][:50], k['launches_per_step'], k['ms_per_step'], k['achieved']) for k in d['roofline']['kernels']]"
timeout 900 python -m pytest tests/test_zz_gpu_fullsize.py -m gpu -x -q -k "sd15 or pair or conv" 2>&1 | tail -3
