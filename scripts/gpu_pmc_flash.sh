# PMC passes on the flash-attention kernel at the SD1.5 64x64-level shape (L = 4096, d = 40, 128 (head, image) pairs): scripts/pmc_flash.py
D=gpurun_out/r03c
mkdir -p $D
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_flash_attn" -d $R/$D -o pmc_$i -- python $R/scripts/pmc_flash.py > $R/$D/run_$i.log 2>&1 )
done
python - <<'PY'
import sqlite3, glob, os
D='gpurun_out/r03c'
for db in sorted(glob.glob(D+'/pmc_*_results.db')):
    con=sqlite3.connect(db)
    try:
        rows=list(con.execute("select counter_name, avg(v), count(*) from (select counter_name, dispatch_id, sum(counter_value) v from pmc_events where name like '%k_flash_attn%' group by counter_name, dispatch_id) group by counter_name"))
        dur=list(con.execute("select avg(end-start), count(*) from kernels where name like '%k_flash_attn%'"))
    except Exception as e:
        rows=[("error "+str(e),0,0)]; dur=[]
    print(os.path.basename(db), dur)
    for r in rows: print("   %-34s %.4e  (%d dispatches)"%r)
PY
tail -3 $D/run_1.log
rm -f $D/*_results.db
