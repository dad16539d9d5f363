#!/bin/bash
# SQ-level PMC passes (three counter sets, separate runs, --kernel-trace only) over one isolated launch: usage gpu_pmc_sq.sh <tag> <kernel regex> <python script + args>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; RE=$2; SCRIPT=$3; shift 3
D=gpurun_out/pmc_$TAG
mkdir -p $D
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$RE" -d $R/$D -o pmc_$i -- python $R/$SCRIPT "$@" > $R/$D/run_$i.log 2>&1 )
done
python - "$D" "$RE" <<'PY' > gpurun_out/pmc_$TAG.txt
import sqlite3, glob, os, sys
D, RE = sys.argv[1], sys.argv[2]
for db in sorted(glob.glob(D+'/pmc_*_results.db')):
    con=sqlite3.connect(db)
    tabs=[r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        kn=[t for t in tabs if t=='kernels'][0]
        names=list(con.execute("select name, avg(end-start), count(*) from kernels group by name"))
        for n in names: print(os.path.basename(db), "kernel %s  avg %.1f us  x%d" % (n[0][:90], n[1]/1e3, n[2]))
        rows=list(con.execute("select name, counter_name, avg(v), count(*) from (select name, counter_name, dispatch_id, sum(counter_value) v from pmc_events group by name, counter_name, dispatch_id) group by name, counter_name"))
        for r in rows: print("   %-60s %-32s %.4e  (%d dispatches)"%(r[0][:60], r[1], r[2], r[3]))
    except Exception as e:
        print("error", e, tabs[:20])
PY
cat gpurun_out/pmc_$TAG.txt | cut -c1-200
tail -n 3 $D/run_1.log; rm -f $D/*_results.db
