#!/bin/bash
# round 4, call M: conv3w wave-priority variants (conv3w_prio 0..3) on the SD1.5 step; k_qgemm16 tile rows (qgemm16_rb) on the FLUX forward; stream-K default policy check
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv2d or stream_k or quantised" ) > gpurun_out/r5m_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5m_tests.log
for p in 1 2 3; do ( SDCPP_BACKEND_OPTS="conv3w_prio=$p" timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x -k "conv" ) > gpurun_out/r5m_tests_prio$p.log 2>&1; echo "rc=$?" >> gpurun_out/r5m_tests_prio$p.log; done
timeout 300 python scripts/ab_bench.py conv3w_prio 0,1,2,3 3 4 > gpurun_out/r5m_ab_conv3w_prio.txt 2>&1
rm -f gpurun_out/r5m_family_flux.txt
for rep in 1 2; do for o in "qgemm16_rb=0" "qgemm16_rb=3" "qgemm16_rb=2"; do echo "#### flux $o" >> gpurun_out/r5m_family_flux.txt; timeout 300 python scripts/family_times.py flux $o 2>&1 | head -6 >> gpurun_out/r5m_family_flux.txt; done; done
( SDCPP_BACKEND_OPTS="qgemm16_rb=3" timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "quantised or flux or clip or t5" ) > gpurun_out/r5m_tests_rb3.log 2>&1; echo "rc=$?" >> gpurun_out/r5m_tests_rb3.log
tail -n 3 gpurun_out/r5m_tests*.log; tail -n 5 gpurun_out/r5m_ab_conv3w_prio.txt; grep "####\|==\|few-row" gpurun_out/r5m_family_flux.txt
