#!/bin/bash
# round 3, call F: ping-pong flash kernel, reads up front + pinned barrier: check + PMC
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 400 python scripts/flash_check.py ) > gpurun_out/r3f_flash_check.log 2>&1
D=gpurun_out/r3f_pmc
mkdir -p $D
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_WAVES"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_flash" -d $R/$D -o pmc_$i -- python $R/scripts/pmc_flash_pp.py > $R/$D/run_$i.log 2>&1 )
done
python - > gpurun_out/r3f_pmc_flash.txt 2>&1 <<'PY'
import sqlite3, glob, os
D='gpurun_out/r3f_pmc'
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
rm -rf $D
cat gpurun_out/r3f_flash_check.log | cut -c1-200
