D=gpurun_out/r02e
mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sdxl > $D/bench.jsonl 2>$D/bench.err; python -c "
import json; d=json.loads(open('$D/bench.jsonl').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved']); [print(k['name'][:50], k['launches_per_step'], k['ms_per_step'], k['achieved']) for k in d['roofline']['kernels']]"
timeout 1200 python -m pytest tests/test_zz_gpu_fullsize.py -m gpu -q -s 2>&1 | grep -E "GPU|PSNR|passed|failed|Error|error|assert|vs oracle" | tee $D/fullwidth_parity.txt | tail -30
