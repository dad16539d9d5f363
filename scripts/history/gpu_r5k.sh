#!/bin/bash
# round 4, call K: stream-K for the pipelined 256 x 256 Linear tile: correctness (new op test, DiT Linear shapes at size, DiT block / model tests), then
# per-family times of the FLUX / SD3.5 forwards with streamk = 0 / 1 (alternating), and the SD1.5 step A/B (must be unchanged: no stream-K launch there)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_zz_gpu_config_shapes.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -q -x -k "stream_k or config_linear or mmdit or flux or sd35 or joint" ) > gpurun_out/r5k_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5k_tests.log
rm -f gpurun_out/r5k_family_dit.txt
for o in "streamk=0" "streamk=1" "streamk=0" "streamk=1"; do
  for m in flux sd35; do echo "#### $m $o" >> gpurun_out/r5k_family_dit.txt; timeout 300 python scripts/family_times.py $m $o 2>&1 | head -5 >> gpurun_out/r5k_family_dit.txt; done
done
timeout 200 python scripts/ab_bench.py streamk 0,1 2 4 > gpurun_out/r5k_ab_sd15.txt 2>&1
tail -n 5 gpurun_out/r5k_tests.log; grep "####\|==\|Linear" gpurun_out/r5k_family_dit.txt; tail -n 2 gpurun_out/r5k_ab_sd15.txt
