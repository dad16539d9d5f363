# round 6, last call r09zz: the whole GPU suite at the final code, smoke, the default bench line, a rocprofv3 kernel summary of the same command
D=gpurun_out/r09zz; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q > $D/suite.log 2>&1; tail -3 $D/suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $D/bench_default.jsonl 2> $D/bench_default.err; tail -c 300 $D/bench_default.jsonl
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $R/$D -o stats -- python $R/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --skip-legs sdxl,flux,sd35,sdxl_b8 > $R/$D/bench_under_rocprof.jsonl 2> $R/$D/stderr.log )
python scripts/rocpd_stats.py $D/stats_results.db $D/kernel_stats.csv | head -8 | cut -c1-170
rm -f $D/*_results.db
