#!/bin/bash
# round 4, call R: SQ-level PMC of the flash-attention kernels at the benchmarked shapes (MFMA utilisation, sustained clock) and HBM bandwidth of the in-register dequant kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
bash scripts/gpu_pmc_sq.sh flash_d40 "k_flash" scripts/pmc_flash_shapes.py 40 4096 128 > /dev/null 2>&1
bash scripts/gpu_pmc_sq.sh flash_d64 "k_flash" scripts/pmc_flash_shapes.py 64 4250 76 > /dev/null 2>&1
bash scripts/gpu_pmc_sq.sh flash_d128 "k_flash" scripts/pmc_flash_shapes.py 128 4352 24 > /dev/null 2>&1
timeout 120 python scripts/qgemv_bench.py > gpurun_out/r5r_qgemv_bench.txt 2>&1
for t in flash_d40 flash_d64 flash_d128; do grep -v "rocclr\|simple_timer" gpurun_out/pmc_$t.txt | cut -c1-150; done; cat gpurun_out/r5r_qgemv_bench.txt
