#!/usr/bin/env python
"""HBM counter traffic per LAUNCH SHAPE: joins the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) with the kernel dispatch table and groups
the dispatches of the selected kernels by (kernel name, grid): launches, average fetch bytes (FETCH_SIZE x 1024 x 2 — the gfx950 correction, calibrated for
4 / 8 / 16-byte-per-lane reads and LDS-DMA alike: profiles/r06b_fetch_calib.json), average write bytes (WRITE_SIZE x 1024, calibrated 1.0), and their sum.
usage: pmc_by_shape.py <fetch.db> <write.db> <kernel substring[|substring...]> [out.txt]"""
import re
import sqlite3
import sys


def cols(con, table):
    return [r[1] for r in con.execute(f"pragma table_info({table})")]


def per_dispatch(db, counter, pats):
    con = sqlite3.connect(db)
    where = " or ".join("p.name like ?" for _ in pats)
    kc = cols(con, "kernels")
    grid = [c for c in ("grid_size_x", "grid_x", "grid_size") if c in kc]
    gy = [c for c in ("grid_size_y", "grid_y") if c in kc]
    wg = [c for c in ("workgroup_size_x", "workgroup_x", "workgroup_size") if c in kc]
    idc = "dispatch_id" if "dispatch_id" in kc else "id"
    sel = ", ".join([f"k.{grid[0]}" if grid else "0", f"k.{gy[0]}" if gy else "1", f"k.{wg[0]}" if wg else "0"])
    q = (f"select p.dispatch_id, p.name, sum(p.counter_value), {sel} from pmc_events p left join kernels k on k.{idc} = p.dispatch_id "
         f"where p.counter_name = ? and ({where}) group by p.dispatch_id order by p.dispatch_id")
    try:
        return list(con.execute(q, (counter, *[f"%{x}%" for x in pats])))
    except sqlite3.OperationalError as e:   # schema without the join columns: names only
        sys.stderr.write(f"[pmc_by_shape] join failed ({e}); grouping by kernel name only; kernels columns: {kc}\n")
        q = f"select p.dispatch_id, p.name, sum(p.counter_value), 0, 1, 0 from pmc_events p where p.counter_name = ? and ({where}) group by p.dispatch_id order by p.dispatch_id"
        return list(con.execute(q, (counter, *[f"%{x}%" for x in pats])))


def main():
    fdb, wdb, pat = sys.argv[1:4]
    out = open(sys.argv[4], "w") if len(sys.argv) > 4 else None
    pats = pat.split("|")
    F = per_dispatch(fdb, "FETCH_SIZE", pats)
    W = per_dispatch(wdb, "WRITE_SIZE", pats)
    # the two passes run the same program: the i-th selected dispatch of one is the i-th of the other
    n = min(len(F), len(W))
    groups = {}
    for i in range(n):
        _, name, fv, gx, gy, wg = F[i]
        wv = W[i][2]
        name = re.sub(r"\s+", " ", name.split("(")[0])
        wgs = (gx // wg if wg else gx) if gx else 0
        key = (name, wgs, gy)
        g = groups.setdefault(key, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += fv * 1024 * 2
        g[2] += wv * 1024
    lines = [f"# {n} dispatches matching {pat!r}; bytes per launch (FETCH_SIZE doubled, WRITE_SIZE as is); workgroups = grid / workgroup size",
             "launches | workgroups x slices | fetch MB | write MB | total MB | kernel"]
    tot = [0.0, 0.0]
    for (name, wgs, gy), (c, f, w) in sorted(groups.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        lines.append(f"{c:5d} | {wgs:6d} x {gy:2d} | {f / c / 1e6:9.1f} | {w / c / 1e6:9.1f} | {(f + w) / c / 1e6:9.1f} | {name[:110]}")
        tot[0] += f
        tot[1] += w
    lines.append(f"TOTAL over {n} launches: fetch {tot[0] / 1e6:.0f} MB, write {tot[1] / 1e6:.0f} MB, per launch {(tot[0] + tot[1]) / max(n, 1) / 1e6:.1f} MB")
    txt = "\n".join(lines)
    print(txt)
    if out:
        out.write(txt + "\n")


if __name__ == "__main__":
    main()
