# round 2, call b: the new full-width oracle parity tests + the whole -m gpu suite, then the default bench line (kernels[] + sdxl + host_loop legs)
D=gpurun_out/r02b
mkdir -p $D
nproc > $D/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $D/nproc.txt
timeout 900 python -m pytest tests/test_zz_gpu_fullsize.py -m gpu -x -q -s -k "oracle" 2>&1 | grep -E "rel-L2|PSNR|passed|failed|Error|error|assert" | tee $D/fullwidth_parity.txt | tail -20
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $D/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; tail -c 6000 $D/bench_default.jsonl; tail -3 $D/bench_default.err
