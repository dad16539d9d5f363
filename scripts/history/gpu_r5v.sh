#!/bin/bash
# round 4, call V: stream-K hybrid (streamk = 3): op test, DiT block tests with it on, FLUX / SD3.5 family A/B against streamk = 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "stream_k" ) > gpurun_out/r5v_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5v_tests.log
( SDCPP_BACKEND_OPTS="streamk=3" timeout 400 python -m pytest tests/test_zz_gpu_config_shapes.py tests/test_zz_gpu_fullsize.py tests/test_gpu_model.py -m gpu -q -x -k "config_linear or flux or mmdit or sd35" ) > gpurun_out/r5v_tests_sk3.log 2>&1; echo "rc=$?" >> gpurun_out/r5v_tests_sk3.log
rm -f gpurun_out/r5v_family.txt
for rep in 1 2; do for o in "streamk=0" "streamk=3"; do for m in flux sd35; do echo "#### $m $o" >> gpurun_out/r5v_family.txt; timeout 300 python scripts/family_times.py $m $o 2>&1 | head -3 >> gpurun_out/r5v_family.txt; done; done; done
MI355X_KTIME_DUMP=gpurun_out/r5v_shapes_flux_sk3.txt timeout 300 python scripts/family_times.py flux streamk=3 > /dev/null 2>&1
tail -n 3 gpurun_out/r5v_tests.log gpurun_out/r5v_tests_sk3.log; grep "####\|==\|Linear" gpurun_out/r5v_family.txt; grep "Linear MFMA" gpurun_out/r5v_shapes_flux_sk3.txt | sort -t"|" -k4 -n -r | head -6
