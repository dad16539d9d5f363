D=gpurun_out/r02s
mkdir -p $D
for o in "" "fgemv=0" "qgemv=0"; do echo "### dist worker pair_a opts=[$o]"; SDCPP_BACKEND_OPTS="$o" timeout 300 python tests/gpu_dist_worker.py pair_a 2>&1 | tail -12; done
timeout 1500 python -m pytest tests -m gpu -q --deselect "tests/test_gpu_dist.py::test_cfg_pair_split_device_side_exchange" 2>&1 | tail -15
timeout 300 python scripts/qgemm16_probe.py > $D/qgemm_paths_probe.txt 2>&1; tail -70 $D/qgemm_paths_probe.txt
for opts in "fgemv=1" "fgemv=0" "fgemv=1"; do
echo "#### $opts"
timeout 300 python scripts/family_times.py sd15 $opts 2>&1 | grep -E "==|Linear|split|unary|rows|few-row" 
done
