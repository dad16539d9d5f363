mkdir -p gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-include-regex "flash" -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o p1 -- python $GRAFT_REPO_ROOT/scripts/pmc_flash.py 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-include-regex "flash" -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o p2 -- python $GRAFT_REPO_ROOT/scripts/pmc_flash.py 2>&1 | tail -3
cd $GRAFT_REPO_ROOT && python - <<'PY'
import sqlite3, collections
for db in ['gpurun_out/pmc2/p1_results.db','gpurun_out/pmc2/p2_results.db']:
    con=sqlite3.connect(db)
    d=collections.defaultdict(lambda: collections.defaultdict(float)); meta={}
    for name, disp, dur, cn, cv in con.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
        d[disp][cn]+=cv; meta[disp]=(name,dur)
    for disp in sorted(d)[-1:]:
        print(db, meta[disp][0][:50], "%.1f us"%(meta[disp][1]/1e3))
        for k,v in sorted(d[disp].items()): print("   %-28s %.4g"%(k,v))
PY
