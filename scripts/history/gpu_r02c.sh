D=gpurun_out/r02c
mkdir -p $D
timeout 600 python scripts/t320_check.py 2>&1 | tee $D/t320_check.txt | cut -c1-400
timeout 1200 python -m pytest tests/test_zz_gpu_fullsize.py -m gpu -q -s -k "oracle" 2>&1 | grep -E "rel-L2|PSNR|passed|failed|Error|error|assert|vs oracle" | tee $D/fullwidth_parity.txt | tail -30
