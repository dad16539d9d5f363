#!/usr/bin/env python
"""First non-finite node of a DiT forward at batch 2 (cuts behind a few nodes, then behind every node of the window).  usage: dit_batch_nan_locate.py <model attr> <wtype attr> <latent> [batch]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

torch.cuda.init()
import sdcpp_amd as sd

mattr, wattr, lat = sys.argv[1], sys.argv[2], int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2
sd.load_mi355x_backend()
rng = np.random.default_rng(99)
flux = "FLUX" in mattr
ntok, cdim, ydim, ch = (256, 4096, 768, 16) if flux else (154, 4096, 2048, 16)
x = rng.standard_normal((n, ch, lat, lat)).astype(np.float32)
t = np.full((n,), 0.5 if flux else 500.0, dtype=np.float32)
ctx = rng.standard_normal((n, ntok, cdim)).astype(np.float32)
y = rng.standard_normal((n, ydim)).astype(np.float32)
eng = sd.Engine(model=getattr(sd, mattr), backend="MI355X0", wtype=getattr(sd, wattr), flash_attn=True)
out = eng.unet_forward(x, t, ctx, y)
print(f"{mattr} latent {lat} batch {n}: whole graph finite {bool(np.isfinite(out).all())}", flush=True)
import ctypes as C
info, ptr2idx = [], {}


def note(i, ts):
    ptr2idx[C.addressof(ts)] = i
    srcs = []
    for k in range(4):
        if ts.src[k]:
            a = C.cast(ts.src[k], C.c_void_p).value
            st = C.cast(ts.src[k], C.POINTER(sd.GgmlTensor)).contents
            srcs.append((ptr2idx.get(a, -1), int(st.op), [int(st.ne[d]) for d in range(4)], [int(st.nb[d]) for d in range(4)], st.name.decode(errors="replace")[:24]))
    info.append((i, int(ts.op), [int(ts.ne[d]) for d in range(4)], [int(ts.nb[d]) for d in range(4)], ts.name.decode(errors="replace")[:28], srcs))
    return False


with sd.EvalTrace(note) as tr0:
    eng.unet_forward(x, t, ctx, y)
nn = tr0.asked
print("nodes", nn, flush=True)
lo, hi = 0, nn - 1
if len(sys.argv) > 5 and sys.argv[5] == "sets":
    for spec in sys.argv[6:]:
        cuts = set(int(v) for v in spec.split(","))
        with sd.EvalTrace(lambda i, ts: i in cuts, with_src1=False) as tr:
            o2 = eng.unet_forward(x, t, ctx, y)
        rec = [(i, None if v is None else bool(np.isfinite(v).all())) for (i, op, name, v, _s) in tr.records]
        for (i, op, name, v, _s) in tr.records:
            if v is not None and not np.isfinite(v).all():
                bad = ~np.isfinite(v)
                print(f"   node {i} shape {v.shape}: non-finite {int(bad.sum())} of {v.size}")
                for ax in range(v.ndim):
                    other = tuple(a for a in range(v.ndim) if a != ax)
                    cnt = bad.sum(axis=other)
                    nz = np.nonzero(cnt)[0]
                    print(f"      axis {ax} (len {v.shape[ax]}): indices with bad values: {len(nz)}; first {nz[:12].tolist()} last {nz[-6:].tolist()}; counts at those {cnt[nz[:6]].tolist()}")
                break
        print(f"cuts {sorted(cuts)}: final finite {bool(np.isfinite(o2).all())}; records (node, finite) {rec}", flush=True)
    sys.exit(0)
if len(sys.argv) > 5 and sys.argv[5] == "single":
    # ONE cut at a time: which cuts make the final result finite (a fusion across that node is the culprit), and where is the recorded node itself bad
    res = []
    for c in range(nn - 1):
        with sd.EvalTrace(lambda i, ts: i == c, with_src1=False) as tr:
            o2 = eng.unet_forward(x, t, ctx, y)
        v = tr.records[0][3] if tr.records else None
        res.append((c, bool(np.isfinite(o2).all()), None if v is None else bool(np.isfinite(v).all())))
    print("cuts that make the final result finite:", [c for c, f, r in res if f])
    print("cuts whose recorded node is non-finite:", [c for c, f, r in res if r is False][:60])
    for (i, op, ne, nb, name, srcs) in info:
        if res[i][1] if i < len(res) else False:
            print(f"node {i:4d} op {op:3d} ne {ne} '{name}'  <- " + " | ".join(f"[{si} op {so} ne {sne} '{sn}']" for si, so, sne, snb, sn in srcs), flush=True)
    sys.exit(0)
if len(sys.argv) > 6:
    lo, hi = int(sys.argv[5]), int(sys.argv[6])
    cuts = set(range(lo, hi + 1))
    with sd.EvalTrace(lambda i, ts: i in cuts, with_src1=False) as tr:
        o2 = eng.unet_forward(x, t, ctx, y)
    for (i, op, name, v, _s) in tr.records:
        if v is None:
            print(f"  node {i} op {op} '{name}': not fetched")
        else:
            fin = np.isfinite(v)
            per = [int((~np.isfinite(v[0, b])).sum()) for b in range(v.shape[1])] if v.ndim == 4 else []
            print(f"  node {i} op {op} '{name}' {v.shape}: non-finite {int((~fin).sum())} of {v.size} (per dim-2 slice {per}), |max| finite {float(np.abs(v[fin]).max()) if fin.any() else float('nan'):.3e}", flush=True)
    sys.exit(0)
for rnd in range(4):
    step = max(1, (hi - lo) // 24)
    cuts = set(range(lo, hi + 1, step)) | {hi}
    with sd.EvalTrace(lambda i, ts: i in cuts, with_src1=False) as tr:
        o2 = eng.unet_forward(x, t, ctx, y)
    bad = None
    prev = lo
    for (i, op, name, v, _s) in tr.records:
        if v is not None and not np.isfinite(v).all():
            bad = (i, op, name, v.shape, int((~np.isfinite(v)).sum()), v.size)
            break
        prev = i
    print(f"round {rnd}: cuts every {step} in [{lo}, {hi}] ({len(tr.records)} records), sliced result finite {bool(np.isfinite(o2).all())}; last finite cut {prev}, first bad {bad}", flush=True)
    if bad is None:
        break
    lo, hi = prev, bad[0]
    if step == 1:
        break

for (i, op, ne, nb, name, srcs) in info[max(0, lo - 14):hi + 2]:
    print(f"node {i:4d} op {op:3d} ne {ne} nb {nb} '{name}'  <- " + " | ".join(f"[{si} op {so} ne {sne} nb {snb} '{sn}']" for si, so, sne, snb, sn in srcs), flush=True)
