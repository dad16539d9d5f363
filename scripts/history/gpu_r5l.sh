#!/bin/bash
# round 4, call L: stream-K with XCD-adjacent unit ranges; GEGLU launches; tile thresholds under stream-K (t256p_min_nt_sk / t256p_min_tiles_sk)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "stream_k or geglu or feed_forward" ) > gpurun_out/r5l_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5l_tests.log
rm -f gpurun_out/r5l_family.txt gpurun_out/r5l_shapes_flux_sk*.txt
for rep in 1 2; do
for o in "streamk=0" "streamk=1" "streamk=1 t256p_min_nt_sk=32" "streamk=1 t256p_min_nt_sk=32 t256p_min_tiles_sk=128"; do
  for m in flux sdxl sd15 sd35; do echo "#### $m $o" >> gpurun_out/r5l_family.txt; timeout 300 python scripts/family_times.py $m $o 2>&1 | head -3 >> gpurun_out/r5l_family.txt; done
done
done
MI355X_KTIME_DUMP=gpurun_out/r5l_shapes_flux_sk0.txt timeout 300 python scripts/family_times.py flux streamk=0 > /dev/null 2>&1
MI355X_KTIME_DUMP=gpurun_out/r5l_shapes_flux_sk1.txt timeout 300 python scripts/family_times.py flux streamk=1 > /dev/null 2>&1
tail -n 3 gpurun_out/r5l_tests.log; grep "####\|==\|Linear" gpurun_out/r5l_family.txt
