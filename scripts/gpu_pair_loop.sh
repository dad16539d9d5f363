#!/bin/bash
# VERDICT r5 task 9: the two-context CFG-pair scenario (two engine contexts on ONE device, two host threads) N times in fresh processes, plain and with
# blocking launches; any death by signal / non-zero exit is counted and its stderr kept.   usage: scripts/gpu_pair_loop.sh [N plain] [N blocking] [out file]
NP=${1:-160}; NB=${2:-40}; OUT=${3:-gpurun_out/pair_loop.txt}
ok=0; bad=0
for i in $(seq 1 $NP); do
  if python tests/gpu_dist_worker.py pair_a > /tmp/pl.out 2> /tmp/pl.err && grep -q OK /tmp/pl.out; then ok=$((ok+1)); else bad=$((bad+1)); echo "--- plain run $i failed" >> $OUT.err; tail -5 /tmp/pl.err >> $OUT.err; fi
done
okb=0; badb=0
for i in $(seq 1 $NB); do
  if HIP_LAUNCH_BLOCKING=1 python tests/gpu_dist_worker.py pair_a > /tmp/pl.out 2> /tmp/pl.err && grep -q OK /tmp/pl.out; then okb=$((okb+1)); else badb=$((badb+1)); echo "--- blocking run $i failed" >> $OUT.err; tail -5 /tmp/pl.err >> $OUT.err; fi
done
echo "two-context pair_a scenario, fresh process each: plain $ok ok / $bad failed of $NP; HIP_LAUNCH_BLOCKING=1 $okb ok / $badb failed of $NB" | tee $OUT
