# in-loop dequantisation: parity test, per-shape probe, FLUX A/B   (usage: gpurun -- 'bash scripts/gpu_qinloop.sh r08d')
D=gpurun_out/$1; mkdir -p $D
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "dequantised_in_the_gemm" 2>&1 | tail -3
(timeout 120 python scripts/qinloop_probe.py 4352 3072 12288 q4; timeout 100 python scripts/qinloop_probe.py 4352 12288 3072 q4; timeout 100 python scripts/qinloop_probe.py 8192 3072 3072 q8) 2>&1 | grep -v amdgpu.ids | tee $D/qinloop_probe.txt
timeout 200 python scripts/ab_leg.py flux qinloop_min_rows 0,513 3 5 2>&1 | grep -v "^round\|amdgpu.ids" > $D/ab_flux_qinloop.txt; head -12 $D/ab_flux_qinloop.txt
