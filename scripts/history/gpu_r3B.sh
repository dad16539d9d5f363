#!/bin/bash
# round 3, call B2: PMC traffic of the dominant conv launch in isolation (no residual), window kernel vs round-2 per-tap kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3B
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_conv3w|k_gemm16" -d $R/gpurun_out/r3B -o pmc_${v}_$c -- python $R/scripts/pmc_conv3w.py conv3w=$v > $R/gpurun_out/r3B/run_${v}_$c.log 2>&1 )
done; done
python - > gpurun_out/r3B_pmc_conv3w_isolated.txt <<'PY'
import sqlite3, glob
for v in (1, 0):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        db = glob.glob(f"gpurun_out/r3B/**/pmc_{v}_{c}_results.db", recursive=True)[0]
        con = sqlite3.connect(db)
        rows = list(con.execute("select name, count(*), avg(v) from (select name, dispatch_id, sum(counter_value) v from pmc_events where counter_name = ? group by name, dispatch_id) group by name", (c,)))
        out[c] = rows
    for (name, n, f), (_, _, w) in zip(out["FETCH_SIZE"], out["WRITE_SIZE"]):
        print(f"conv3w={v}  {name[:70]:70s} launches {n}  fetch {f*1024*2/1e6:8.1f} MB (doubled)  raw {f*1024/1e6:8.1f} MB   write {w*1024/1e6:8.1f} MB")
print("NHWC f16 image 41.9 MB, weight image 1.8 MB, f32 output 83.9 MB")
PY
rm -rf gpurun_out/r3B
cat gpurun_out/r3B_pmc_conv3w_isolated.txt
