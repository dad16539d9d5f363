# round 6, call r09i: activation folded into the conv operand image (fuse_act_pack): parity / bit-identity tests, TAESD kernel table, SD1.5 step A/B (the option must be
# neutral there: no UNet node takes the new path, the pack kernels gained an activation code and an optional f32 write-back)
D=gpurun_out/r09i; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ref_graphs.py tests/test_gpu_graph_views.py -m gpu -q -x -s -k "taesd or TAE or activation_folded or vae_decode or views" > $D/tests.log 2>&1; tail -3 $D/tests.log
grep -E "TAESD|kernels unfused" $D/tests.log | cut -c1-200
for a in "SD15_TINY 64 1" "SD15_TINY 64 8" "SD35_TINY 128 1"; do timeout 200 python scripts/tae_probe.py $a 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $D/tae_probe.txt
timeout 400 python scripts/ab_bench.py fuse_act_pack 0,1 3 6 2>&1 | grep -v "amdgpu.ids" | tail -8 | tee $D/ab_sd15.txt
