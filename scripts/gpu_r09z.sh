# round 6, call r09z: the whole GPU suite at the final code, then the end-of-round measurement set (smoke, default bench line, rocprofv3 kernel summary, PMC traffic)
D=gpurun_out/r09z; mkdir -p $D
timeout 1700 python -m pytest tests -m gpu -x -q > $D/suite.log 2>&1; tail -3 $D/suite.log
D=$D bash scripts/gpu_round_end3.sh > $D/round_end.log 2>&1; tail -30 $D/round_end.log | cut -c1-400
