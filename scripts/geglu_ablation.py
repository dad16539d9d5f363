#!/usr/bin/env python
"""Where the time of the K = 320 .. 1280 Linears of a SpatialTransformer goes (EXPERIMENTS build: SDCPP_BACKEND_LIB=.../lib_exp/libggml-mi355x.so).
FeedForward = FF1 (GEGLU epilogue, f16 rows out) + FF2 (+bias +residual); timing ablations of the FF1 launch through option gemm16_abl:
5 = epilogue without the GELU arithmetic, 6 = epilogue without stores, 7 = one k-step instead of the main loop.  HIP events per dispatch;
the FF2 launch is the same in every column (its time is printed separately from the per-shape dump)."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F16, F32, Graph, tensor_struct

sd.load_mi355x_backend()
L = sd.lib()
rng = np.random.default_rng(0)
REPS = 5


def nb(t):
    s = tensor_struct(t)
    return [int(s.nb[i]) for i in range(4)]


def case(tokens, dim):
    inner = 4 * dim
    x = rng.standard_normal((1, tokens, dim)).astype(np.float32)
    w1 = (rng.standard_normal((2 * inner, dim)) / np.sqrt(dim)).astype(np.float32)
    b1 = rng.standard_normal(2 * inner).astype(np.float32)
    w2 = (rng.standard_normal((dim, inner)) / np.sqrt(inner)).astype(np.float32)
    b2 = rng.standard_normal(dim).astype(np.float32)

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

    line = f"tokens {tokens} dim {dim} (FF1 {2.0 * tokens * dim * 2 * inner / 1e9:.1f} GF, FF2 {2.0 * tokens * dim * inner / 1e9:.1f} GF):"
    for abl in (0, 5, 6, 7):
        sd.backend_set_option("gemm16_abl", abl)
        with Graph("MI355X0") as g:
            node = build(g)
            g.run(node)
            gf = L.ggml_new_graph_custom(g.ctx, 256, False)
            L.ggml_build_forward_expand(gf, node)
            sd.kernel_timing_enable(1 << 2)
            for _ in range(REPS):
                L.ggml_backend_graph_compute(g.backend, gf)
            t = sd.kernel_timings()
            sd.kernel_timing_enable(0)
        ms = sum(f["total_ms"] for f in t) / REPS
        line += f"  abl {abl}: FF1+FF2 {ms * 1e3:7.1f} us"
    sd.backend_set_option("gemm16_abl", 0)
    print(line, flush=True)


if __name__ == "__main__":
    case(65536, 320)
    case(16384, 640)
    case(4096, 1280)
