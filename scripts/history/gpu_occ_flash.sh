D=gpurun_out/r03d
mkdir -p $D
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE SQ_CYCLES" "SQ_WAVE_CYCLES SQ_WAVES"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_flash_attn" -d $R/$D -o occ_$i -- python $R/scripts/pmc_flash.py > $R/$D/run_$i.log 2>&1 ); echo "set $i rc=$?"
done
python - <<'PY'
import sqlite3, glob, os
D='gpurun_out/r03d'
for db in sorted(glob.glob(D+'/occ_*_results.db')):
    con=sqlite3.connect(db)
    try:
        rows=list(con.execute("select counter_name, avg(v), count(*) from (select counter_name, dispatch_id, sum(counter_value) v from pmc_events where name like '%k_flash_attn%' group by counter_name, dispatch_id) group by counter_name"))
        dur=list(con.execute("select avg(duration), count(*), max(grid_x), max(grid_y), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count) from kernels where name like '%k_flash_attn%'"))
    except Exception as e:
        rows=[("error "+str(e),0,0)]; dur=[]
    print(os.path.basename(db), dur)
    for r in rows: print("   %-34s %.4e  (%d dispatches)"%r)
PY
grep -i "error\|invalid\|not found" $D/run_*.log | head -5
rm -f $D/*_results.db
