#!/bin/bash
# round 3, call C: ping-pong flash kernel (check + PMC), conv3w with 16-wide maps and the tuned split rule, native RCCL exchange + file -> GPU loader tests, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 400 python scripts/flash_check.py ) > gpurun_out/r3c_flash_check.log 2>&1
( time timeout 500 python scripts/conv3w_check.py ) > gpurun_out/r3c_conv3w_check.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_model_io.py tests/test_gpu_dist.py tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_zz_gpu_config_shapes.py -m gpu -q -x \
    -k "not vae_mid" ) > gpurun_out/r3c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3c_tests.log
D=gpurun_out/r3c_pmc
mkdir -p $D
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_WAVES"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_flash" -d $R/$D -o pmc_$i -- python $R/scripts/pmc_flash_pp.py > $R/$D/run_$i.log 2>&1 )
done
python - > gpurun_out/r3c_pmc_flash.txt 2>&1 <<'PY'
import sqlite3, glob, os
D='gpurun_out/r3c_pmc'
for db in sorted(glob.glob(D+'/**/pmc_*_results.db', recursive=True)):
    con=sqlite3.connect(db)
    try:
        rows=list(con.execute("select counter_name, avg(v), count(*) from (select counter_name, dispatch_id, sum(counter_value) v from pmc_events where name like '%k_flash%' group by counter_name, dispatch_id) group by counter_name"))
        dur=list(con.execute("select name, avg(end-start), count(*) from kernels where name like '%k_flash%' group by name"))
    except Exception as e:
        rows=[("error "+str(e),0,0)]; dur=[]
    print(os.path.basename(db), [(d[0][:60], d[1], d[2]) for d in dur])
    for r in rows: print("   %-34s %.4e  (%d dispatches)"%r)
PY
tail -3 $D/run_1.log >> gpurun_out/r3c_pmc_flash.txt
rm -rf $D
( time timeout 300 python bench.py --skip-legs flux,sd35 --no-cpu-baseline ) > gpurun_out/r3c_bench.log 2>&1
tail -2 gpurun_out/r3c_flash_check.log; tail -2 gpurun_out/r3c_conv3w_check.log; tail -4 gpurun_out/r3c_tests.log; tail -c 400 gpurun_out/r3c_bench.log
