# kernel-name sequences of one forward at two latent sizes (rocprofv3 --kernel-trace), to diff the plans
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; D=gpurun_out/r08k; mkdir -p $D
for lat in 32 64; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$D -o tr_$lat -- python $R/scripts/dit_one_forward.py SD35_WIDE2 BF16 $lat 2 > $R/$D/tr_$lat.log 2>&1 ); tail -1 $D/tr_$lat.log
  python - $D/tr_${lat}_results.db $D/seq_$lat.txt <<'PY'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
rows = list(con.execute("select name, start, end from kernels order by start"))
with open(sys.argv[2], "w") as f:
    for n, s, e in rows:
        f.write(f"{(e - s) / 1e3:9.1f} us  {n[:160]}\n")
print(len(rows), "kernels")
PY
  rm -f $D/tr_${lat}_results.db
done
