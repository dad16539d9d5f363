D=gpurun_out/r02x
mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
for opts in "fuse_siblings=1" "fuse_siblings=0" "fuse_siblings=1" "fuse_siblings=0"; do
echo "#### sd15 $opts"
timeout 300 python scripts/family_times.py sd15 $opts 2>&1 | grep -E "==|Linear" 
done
for opts in "fuse_siblings=1" "fuse_siblings=0"; do
echo "#### sdxl $opts"
timeout 300 python scripts/family_times.py sdxl $opts 2>&1 | grep -E "==|Linear" 
done
