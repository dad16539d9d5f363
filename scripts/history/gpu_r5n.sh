#!/bin/bash
# round 4, call N: column-range epilogue (FLUX linear1 writes gelu(mlp) as f16 into linear2's operand image): FLUX tests, family A/B (fuse_split_gelu 0 / 1);
# stream-K default policy (two rounds or more) on FLUX; new conv3w / qgemm16 defaults: conv + quantised tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x -k "flux or stream_k or conv2d or quantised or config_linear" ) > gpurun_out/r5n_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5n_tests.log
rm -f gpurun_out/r5n_family_flux.txt
for rep in 1 2; do for o in "fuse_split_gelu=0 streamk=0" "fuse_split_gelu=1 streamk=0" "fuse_split_gelu=1 streamk=1"; do echo "#### flux $o" >> gpurun_out/r5n_family_flux.txt; timeout 300 python scripts/family_times.py flux $o 2>&1 | head -9 >> gpurun_out/r5n_family_flux.txt; done; done
tail -n 4 gpurun_out/r5n_tests.log; grep "####\|==\|Linear\|f32 rows" gpurun_out/r5n_family_flux.txt
