# round 6, call r09g: the new GPU tests (TAESD, the widened samplers, TAESD through the reference runner), a TAESD kernel table, the default bench line
D=gpurun_out/r09g; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ref_graphs.py -m gpu -q -x -s -k "taesd or more_samplers or TAE" > $D/new_tests.log 2>&1; tail -4 $D/new_tests.log
grep -E "TAESD|trajectory rel-L2|fusion counters" $D/new_tests.log | cut -c1-260
for a in "SD15_TINY 64 1" "SD15_TINY 64 8" "SD35_TINY 128 1"; do timeout 200 python scripts/tae_probe.py $a 2>&1 | grep -v Warning; done | tee $D/tae_probe.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; tail -c 1500 $D/bench_default.jsonl | head -c 700
