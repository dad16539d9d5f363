D=gpurun_out/r02r
mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python scripts/qgemm16_probe.py > $D/qgemm16_probe.txt 2>&1; tail -70 $D/qgemm16_probe.txt
for opts in "fgemv=1 splitk_inkernel=0" "fgemv=0 splitk_inkernel=0" "fgemv=1 splitk_inkernel=1" "fgemv=1 splitk_inkernel=0"; do
echo "#### $opts"
timeout 300 python scripts/family_times.py sd15 $opts 2>&1 | grep -E "==|Linear|split|unary|rows|few-row|conv" 
done
