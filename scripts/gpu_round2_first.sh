# First GPU call of the next round: the measurements round 1 built but could not time (GPU budget exhausted).
#   1. host-driven step vs device-resident trajectory, SD1.5 512^2 batch 8
#   2. end-to-end image with and without the device-resident sampler
#   3. text-encoder latency at real width for SD1.5 and SDXL (f3)
D=gpurun_out/r02a
mkdir -p $D
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline                   | tee $D/bench_host_loop.jsonl | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --device-sampler  | tee $D/bench_device_sampler.jsonl | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --device-sampler --backend-opt pinned_uploads=1 | tee $D/bench_device_sampler_pinned.jsonl | cut -c1-400
SDCPP_BACKEND_OPTS=pinned_uploads=1 timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --e2e              | tee $D/e2e_host_loop.jsonl | cut -c1-400
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --e2e --device-sampler | tee $D/e2e_device_sampler.jsonl | cut -c1-400
timeout 300 python scripts/te_bench.py sd15 sdxl | tee $D/te_bench.jsonl
#   4. explicit LDS-read / MFMA interleave in the 256-row gemm16 tiles (built blind at the end of round 1: parity first, then time)
SDCPP_BACKEND_OPTS=gemm16_sched=3 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
for v in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --backend-opt gemm16_sched=$v | tee $D/bench_sched$v.jsonl | cut -c1-400; done
#   5. which pipeline bounds the dominant kernel: DMA-only / compute-only ablations (wrong results by design, timing only)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$D -o ablation -- python $GRAFT_REPO_ROOT/scripts/gemm_ablation.py > $GRAFT_REPO_ROOT/$D/ablation_wall.txt 2> $GRAFT_REPO_ROOT/$D/ablation.log )
python scripts/rocpd_stats.py $D/ablation_results.db $D/ablation_kernel_stats.csv | grep k_gemm16 | cut -c1-160
#   6. A operand global -> VGPR (k_gemm16d, all tile shapes; built blind at the end of round 1): parity first, then time
SDCPP_BACKEND_OPTS=gemm16_adirect=1 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --backend-opt gemm16_adirect=1 | tee $D/bench_adirect.jsonl | cut -c1-400
#   7. two K slices for the 193..384-workgroup launches (16x16 UNet level)
SDCPP_BACKEND_OPTS=splitk_mid=1 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_zz_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --backend-opt splitk_mid=1 | tee $D/bench_splitk_mid.jsonl | cut -c1-400
#   8. flash attention d = 40: where the 380 us go (wrong-result ablations, timing only)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$D -o flash_ablation -- python $GRAFT_REPO_ROOT/scripts/flash_ablation.py > $GRAFT_REPO_ROOT/$D/flash_ablation_wall.txt 2> $GRAFT_REPO_ROOT/$D/flash_ablation.log )
python scripts/rocpd_stats.py $D/flash_ablation_results.db $D/flash_ablation_kernel_stats.csv | grep k_flash | cut -c1-160
