#!/usr/bin/env python
"""Reduce the two rocprofv3 PMC passes over scripts/fetch_calib.hip: counter value per launch / known bytes per launch, per access width.
usage: fetch_calib.py <fetch.db> <write.db> <bytes per launch> <out.json>
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1024 B (request count x 64 B / 1024)."""
import json
import sqlite3
import sys

fetch_db, write_db, nbytes, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view') and name like 'pmc_events%'")]
    res = {}
    for t in tabs[:1] or ["pmc_events"]:
        for name, disp, val in con.execute(f"select name, dispatch_id, sum(counter_value) from {t} where counter_name = ? group by name, dispatch_id", (counter,)):
            res.setdefault(name.split("(")[0], []).append(val)
    return {k: sum(v) / len(v) for k, v in res.items()}, {k: len(v) for k, v in res.items()}


f, nf = per_kernel(fetch_db, "FETCH_SIZE")
w, nw = per_kernel(write_db, "WRITE_SIZE")
rows = {}
for k in sorted(set(f) | set(w)):
    reads = 0 if "write" in k else nbytes
    writes = 0 if "read" in k else nbytes
    rows[k] = {"launches": nf.get(k, nw.get(k)), "known_read_bytes": reads, "known_write_bytes": writes,
               "FETCH_SIZE_x1024": None if k not in f else f[k] * 1024, "WRITE_SIZE_x1024": None if k not in w else w[k] * 1024,
               "fetch_reported_over_known": None if not reads or k not in f else round(f[k] * 1024 / reads, 4),
               "write_reported_over_known": None if not writes or k not in w else round(w[k] * 1024 / writes, 4)}
json.dump({"bytes_per_launch": nbytes, "kernels": rows,
           "reading": "reported / known = 0.5 means the counter tallies this width's requests at half their size (multiply by 2), 1.0 means it is exact"}, open(out, "w"), indent=1)
for k, r in rows.items():
    print(f"{k:40s} fetch/known {r['fetch_reported_over_known']}  write/known {r['write_reported_over_known']}")
