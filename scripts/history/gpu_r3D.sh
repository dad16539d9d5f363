#!/bin/bash
# round 3, call D2: hipGraph replay (whole-plan graphs; segmented around the timed family's launches while kernel timing is on): tests + bench on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( SDCPP_BACKEND_OPTS=hip_graph=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_zz_gpu_pixels.py -m gpu -q -x ) > gpurun_out/r3D_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3D_tests.log
for hg in 0 1 0 1; do timeout 300 python bench.py --hip-graph $hg --no-cpu-baseline --no-kernels --no-e2e 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('hip_graph', d['config'].get('hip_graph'), 'sd15', d['ms_per_step'], 'conv us', r.get('avg_launch_us'), 'frac', r.get('frac'), 'launches', r.get('launches'), 'sdxl', d['sdxl']['ms_per_step'], 'flux', d['flux']['ms_per_step'], 'sd35', d['sd35']['ms_per_step'])"; done > gpurun_out/r3D_bench_hip_graph.txt 2>&1
tail -4 gpurun_out/r3D_tests.log; cat gpurun_out/r3D_bench_hip_graph.txt
