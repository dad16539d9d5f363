# A/B: non-temporal conv output stores / residual loads (build variant nt) against the default library: step time, dominant kernel, PMC traffic
D=gpurun_out/r02nt
mkdir -p $D
R=$GRAFT_REPO_ROOT
NT=$R/stable-diffusion.cpp_amd/lib_nt/libggml-mi355x.so
for rep in 1 2; do
for v in default nt; do
  LIBV=""; [ $v = nt ] && LIBV=$NT
  SDCPP_BACKEND_LIB=$LIBV timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sdxl --no-e2e --no-kernels 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$v', d['ms_per_step'], 'ms/step', d['value'], 'it/s | conv-256', r['achieved'], 'TF', r['avg_launch_us'], 'us')"
done
done
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && SDCPP_BACKEND_LIB=$NT timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_gemm16" -d $R/$D -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sdxl --no-e2e --no-kernels > /dev/null 2> $R/$D/pmc_$c.log )
done
python scripts/pmc_traffic.py $D/pmc_FETCH_SIZE_results.db $D/pmc_WRITE_SIZE_results.db "k_gemm16<256, 320, true" $D/pmc_traffic_t320_conv_nt.json | grep -E "bytes_per_launch|launches"
rm -f $D/*_results.db
