D=gpurun_out/r02l
mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $D/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $D/bench.jsonl 2>$D/bench.err; python -c "
import json; d=json.loads(open('$D/bench.jsonl').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['e2e'], d['host_loop'], d['sdxl']); [print(k['name'][:50], k['launches_per_step'], k['ms_per_step'], k['achieved'], k['frac']) for k in d['roofline']['kernels']]"
