D=gpurun_out/r02t
mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15
for opts in "fgemv=1" "fgemv=0" "fgemv=1" "fgemv=0"; do
echo "#### $opts"
timeout 300 python scripts/family_times.py sd15 $opts 2>&1 | grep -E "==|few-row" 
done
timeout 300 python scripts/family_times.py sdxl 2>&1 | grep -E "==|few-row|Linear|unary"
timeout 300 python scripts/family_times.py sdxl fgemv=0 qgemv_max_rows=2 2>&1 | grep -E "==|few-row|Linear|unary"
