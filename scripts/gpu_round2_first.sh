# First GPU call of the next round: the measurements round 1 built but could not time (GPU budget exhausted).
#   1. host-driven step vs device-resident trajectory, SD1.5 512^2 batch 8
#   2. end-to-end image with and without the device-resident sampler
#   3. text-encoder latency at real width for SD1.5 and SDXL (f3)
D=gpurun_out/r02a
mkdir -p $D
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline                   | tee $D/bench_host_loop.jsonl | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --device-sampler  | tee $D/bench_device_sampler.jsonl | cut -c1-400
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --e2e              | tee $D/e2e_host_loop.jsonl | cut -c1-400
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --e2e --device-sampler | tee $D/e2e_device_sampler.jsonl | cut -c1-400
timeout 300 python scripts/te_bench.py sd15 sdxl | tee $D/te_bench.jsonl
