D=gpurun_out/r02i
mkdir -p $D
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for abl in 0 2; do
  LIBV=""; [ $abl != 0 ] && LIBV="$R/stable-diffusion.cpp_amd/lib_exp/libggml-mi355x.so"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    ( cd /tmp && SDCPP_BACKEND_LIB=$LIBV timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_gemm16" -d $R/$D -o pmc_${abl}_$tag -- python $R/scripts/pmc_t320.py $abl > /dev/null 2> $R/$D/pmc_${abl}_$tag.log )
  done
done
ls $D
python - <<'PY'
import sqlite3, glob, os
D='gpurun_out/r02i'
for db in sorted(glob.glob(D+'/pmc_*_results.db')):
    con=sqlite3.connect(db)
    try:
        rows=list(con.execute("select counter_name, avg(v), count(*) from (select counter_name, dispatch_id, sum(counter_value) v from pmc_events where name like '%k_gemm16%' group by counter_name, dispatch_id) group by counter_name"))
    except Exception as e:
        rows=[("error "+str(e),0,0)]
    print(os.path.basename(db))
    for r in rows: print("   %-34s %.4e  (%d dispatches)"%r)
PY
