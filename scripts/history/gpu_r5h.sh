#!/bin/bash
# round 4, call H: per-family and per-shape kernel times of the SDXL / FLUX / SD3.5 forwards with the current defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for m in sdxl flux sd35; do
  rm -f gpurun_out/r5h_shapes_$m.txt
  MI355X_KTIME_DUMP=gpurun_out/r5h_shapes_$m.txt timeout 300 python scripts/family_times.py $m > gpurun_out/r5h_family_$m.txt 2>&1
done
head -20 gpurun_out/r5h_family_*.txt
