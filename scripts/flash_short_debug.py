#!/usr/bin/env python
"""Where does k_flash_short differ from the exact softmax?  Full float64 reference on a small case, error map per query row / head / column and
a few hypotheses (keys dropped, rows permuted inside a 32-row block, normalisation)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F16, Graph
sd.load_mi355x_backend()
L = sd.lib()
rng = np.random.default_rng(0)

def ref_attn(q, k16, v16, sc, keys=None):
    s = np.einsum("hqd,hkd->hqk", q.astype(np.float64), k16) * sc
    if keys is not None:
        s = s[:, :, keys]; v16 = v16[:, keys]
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    return np.einsum("hqk,hkd->hqd", p, v16)

def run(d, Lq, Lk, HN):
    q = rng.standard_normal((HN, Lq, d)).astype(np.float32)
    k = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    v = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    sc = 1.0 / np.sqrt(d)
    outs = {}
    for name, val in (("base", 0), ("short", 1)):
        sd.backend_set_option("flash_short", val)
        with Graph("MI355X0") as g:
            node = L.ggml_flash_attn_ext(g.ctx, g.input(q), g.input(k, F16), g.input(v, F16), None, sc, 0.0, 0.0)
            outs[name] = g.run(node)[0].transpose(1, 0, 2)  # [HN, Lq, d]
    sd.backend_set_option("flash_short", 0)
    k16, v16 = k.astype(np.float16).astype(np.float64), v.astype(np.float16).astype(np.float64)
    ref = ref_attn(q, k16, v16, sc)
    print(f"--- d={d} Lq={Lq} Lk={Lk} HN={HN}: base err {np.abs(outs['base']-ref).max():.2e}  short err {np.abs(outs['short']-ref).max():.2e}")
    o = outs["short"]
    err = np.abs(o - ref)
    for h in range(min(HN, 2)):
        rows = err[h].max(-1)
        print(f" head {h}: rows bad: " + "".join("X" if e > 3e-3 else "." for e in rows))
        print(f" head {h}: cols max err: " + " ".join(f"{e:.0e}" for e in err[h].max(0)))
    for name, keys in (("first 64 keys only", np.arange(min(64, Lk))), ("first 32", np.arange(32)), ("keys 64..", np.arange(64, Lk)), ("keys 32..", np.arange(32, Lk))):
        print(f"  hyp {name}: {np.abs(o - ref_attn(q, k16, v16, sc, keys)).max():.2e}")
    # rows permuted inside a block?  best-matching reference row for a few output rows
    for h, i in ((0, 0), (0, 5), (0, 33), (0, Lq - 1), (HN - 1, 40)):
        dist = np.abs(ref[h] - o[h, i]).max(-1)
        j = int(dist.argmin())
        print(f"  out[h={h}, q={i}] best matches ref row {j} (err {dist[j]:.1e}); own-row err {dist[i]:.1e}; |out| {np.abs(o[h,i]).max():.2f} |ref| {np.abs(ref[h,i]).max():.2f} nan {np.isnan(o[h,i]).any()}")
    # unnormalised / scaled?
    ratio = (o * ref).sum(-1) / np.maximum((ref * ref).sum(-1), 1e-30)
    print("  per-row least-squares scale out/ref, head 0, first 40 rows: " + " ".join(f"{r:.2f}" for r in ratio[0, :40]))

run(40, 256, 77, 2)
run(64, 96, 96, 1)
run(40, 130, 77, 8)
