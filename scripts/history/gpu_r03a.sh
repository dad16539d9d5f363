D=gpurun_out/r03a
mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
for opts in "hoist_kv=1" "hoist_kv=0" "hoist_kv=1" "hoist_kv=0"; do
echo "#### sd15 $opts"
timeout 300 python scripts/family_times.py sd15 $opts 2>&1 | grep -E "==|Linear MFMA|conv implicit-GEMM, 256" 
done
for opts in "hoist_kv=1" "hoist_kv=0"; do
echo "#### sdxl $opts"
timeout 300 python scripts/family_times.py sdxl $opts 2>&1 | grep -E "==|Linear MFMA" 
done
