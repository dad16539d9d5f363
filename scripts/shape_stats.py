#!/usr/bin/env python
"""Join a GGML_MI355X_TRACE=1 launch log (stderr of the traced run) with the rocprofv3 kernel trace of the same run:
the i-th 'G16 ...' line is the i-th k_gemm16 dispatch.  Prints per-shape call count, average time and algorithmic TFLOP/s.
usage: shape_stats.py <results.db> <stderr log>"""
import collections
import re
import sqlite3
import sys

db, log = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
kern = [(n, (e - s) / 1e3) for n, s, e in con.execute("select name, start, end from kernels where name like '%k_gemm16%' order by start")]
lines = [l.strip() for l in open(log, errors="ignore") if l.startswith("G16 ") and not l.startswith("G16 tile")]  # ("G16 tile": the tile-choice line of the same launch)
print(f"{len(kern)} dispatches, {len(lines)} trace lines")
n = min(len(kern), len(lines))
agg = collections.OrderedDict()
for (name, us), l in zip(kern[:n], lines[:n]):
    m = dict(re.findall(r"(\w+)=(\S+)", l))
    kind = l.split()[1]
    tmpl = re.search(r"k_gemm16<([^>]*)>", name).group(1).replace(" ", "")
    key = (kind, l[4:], tmpl)
    fl = 2.0 * int(m["rows"]) * int(m["K"]) * int(m["M"])
    a = agg.setdefault(key, [0, 0.0, fl])
    a[0] += 1
    a[1] += us
tot = sum(a[1] for a in agg.values())
print(f"{'shape':95s} {'tmpl':16s} {'n':>4s} {'avg us':>9s} {'TF/s':>7s} {'% g16':>6s}")
for (kind, desc, tmpl), (cnt, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{desc:95s} {tmpl:16s} {cnt:4d} {us/cnt:9.1f} {fl/(us/cnt)/1e6:7.1f} {100*us/tot:6.2f}")
print(f"total gemm16 time {tot/1e3:.2f} ms")
