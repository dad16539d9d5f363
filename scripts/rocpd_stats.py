#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max / share.
usage: python scripts/rocpd_stats.py <results.db> [out.csv]"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes"]
    for n, c, t, a, mn, mx, vg, ag, lds in rows:
        n = re.sub(r"\s+", " ", n).replace(",", ";")
        lines.append(f"{n},{c},{t/1e6:.3f},{a/1e3:.1f},{mn/1e3:.1f},{mx/1e3:.1f},{100*t/tot:.2f},{vg},{ag},{lds}")
    lines.append(f"TOTAL,{sum(r[1] for r in rows)},{tot/1e6:.3f},,,,100,,,")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
