#!/usr/bin/env python
"""FeedForward (FF1 + GEGLU -> FF2 + residual) of a SpatialTransformer block in isolation, per launch (HIP events): usage ff_probe.py tokens dim [key=int ...] [-- key=int ...]
Each `--`-separated option set is timed in turn (alternating, 2 rounds)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F16, F32, Q8_0, Graph, tensor_struct

sd.load_mi355x_backend()
L = sd.lib()
rng = np.random.default_rng(0)
tokens, dim = int(sys.argv[1]), int(sys.argv[2])
sets, cur = [], []
for a in sys.argv[3:]:
    if a == "--":
        sets.append(cur)
        cur = []
    else:
        cur.append((a.split("=")[0], int(a.split("=")[1])))
sets.append(cur)
inner = 4 * dim
x = rng.standard_normal((1, tokens, dim)).astype(np.float32)
w1 = (rng.standard_normal((2 * inner, dim)) / np.sqrt(dim)).astype(np.float32)
b1 = rng.standard_normal(2 * inner).astype(np.float32)
w2 = (rng.standard_normal((dim, inner)) / np.sqrt(inner)).astype(np.float32)
b2 = rng.standard_normal(dim).astype(np.float32)


def nb(t):
    s = tensor_struct(t)
    return [int(s.nb[i]) for i in range(4)]


def build(g):
    xin = g.input(x)
    h = L.ggml_mul_mat(g.ctx, g.weight(w1, F16), xin)
    h = L.ggml_add_inplace(g.ctx, h, g.weight(b1, F32))
    ts = nb(h)
    lo = L.ggml_view_4d(g.ctx, h, inner, tokens, 1, 1, ts[1], ts[2], ts[3], 0)
    hi = L.ggml_view_4d(g.ctx, h, inner, tokens, 1, 1, ts[1], ts[2], ts[3], inner * 4)
    gate = L.ggml_gelu_inplace(g.ctx, L.ggml_cont(g.ctx, hi))
    h = L.ggml_mul(g.ctx, lo, gate)
    y = L.ggml_mul_mat(g.ctx, g.weight(w2, F16), h)
    y = L.ggml_add_inplace(g.ctx, y, g.weight(b2, F32))
    return L.ggml_add(g.ctx, y, xin)


for rnd in range(2):
    for opts in sets:
        for k, v in opts:
            sd.backend_set_option(k, v)
        with Graph("MI355X0") as g:
            node = build(g)
            g.run(node)
            gf = L.ggml_new_graph_custom(g.ctx, 256, False)
            L.ggml_build_forward_expand(gf, node)
            sd.kernel_timing_enable(sd.KF_ALL)
            for _ in range(5):
                L.ggml_backend_graph_compute(g.backend, gf)
            t = sd.kernel_timings()
            sd.kernel_timing_enable(0)
        parts = ", ".join(f"{f['kernel'][:28]} {f['total_ms'] / 5 * 1e3:7.1f} us x{f['launches'] // 5}" for f in sorted(t, key=lambda f: -f["total_ms"]))
        print(f"tokens {tokens} dim {dim} {opts}: total {sum(f['total_ms'] for f in t) / 5 * 1e3:7.1f} us | {parts}", flush=True)
        for k, v in opts:  # back to defaults is the caller's business: options listed in every set
            pass
