# round 6, the very last call: the whole GPU suite on the final code (skip-layer guidance included), smoke, a short default bench line
D=gpurun_out/r09end; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q > $D/suite.log 2>&1; tail -3 $D/suite.log
grep -E "SLG" $D/suite.log | head -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; tail -c 300 $D/bench_default.jsonl
