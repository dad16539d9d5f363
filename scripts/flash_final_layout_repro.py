#!/usr/bin/env python
"""FLASH_ATTN_EXT -> VIEW [d,H,Lq,N] -> CONT (ggml_extend.hpp:1446-1455): the kernel writes the final [C, Lq, N] layout itself.  Finite / exact at several (H, N, L)?"""
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
rng = np.random.default_rng(5)


def exact(q, k, v, scale):
    s = np.einsum("hqd,hkd->hqk", q.astype(np.float64), k.astype(np.float64)) * scale
    s -= s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return np.einsum("hqk,hkd->hqd", p, v.astype(np.float64))


def case(d, H, N, Lq, Lk, fused=True, reshape=False):
    HN = H * N
    q = rng.standard_normal((HN, Lq, d)).astype(np.float32)
    k = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    v = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    sd.backend_set_option("fusion", 1 if fused else 0)
    with Graph("MI355X0") as g:
        o = L.ggml_flash_attn_ext(g.ctx, g.input(q), g.input(k, F16), g.input(v, F16), None, scale, 0.0, 0.0)   # [d, HN, Lq, 1]
        L.ggml_flash_attn_ext_set_prec(o, 10)
        ts = tensor_struct(o)
        vw = L.ggml_view_4d(g.ctx, o, d, H, Lq, N, int(ts.nb[1]), int(ts.nb[2]), int(ts.nb[1]) * H, 0)
        ct = L.ggml_cont(g.ctx, vw)   # [d, H, Lq, N]
        if reshape:
            ct = L.ggml_reshape_3d(g.ctx, ct, d * H, Lq, N)
        out = g.run(ct)
    sd.backend_set_option("fusion", 1)
    ex = exact(q, k.astype(np.float16).astype(np.float32), v.astype(np.float16).astype(np.float32), scale).reshape(N, H, Lq, d).transpose(0, 2, 1, 3)   # [N, Lq, H, d]
    got = np.asarray(out).reshape(N, Lq, H, d)
    fin = np.isfinite(got)
    err = float(np.linalg.norm(np.where(fin, got, 0) - ex) / np.linalg.norm(ex))
    per = [(int((~fin[n]).sum()), float(np.linalg.norm(np.where(fin[n], got[n], 0) - ex[n]) / np.linalg.norm(ex[n]))) for n in range(N)]
    print(f"d {d} H {H} N {N} Lq {Lq} Lk {Lk} fused {fused} reshape {reshape}: non-finite {int((~fin).sum())} of {got.size}, rel-L2 {err:.2e}, per image (non-finite, rel-L2) {per}", flush=True)


for a in ((64, 38, 2, 1178, 1178, True, True), (64, 38, 1, 1178, 1178, True, True), (64, 38, 2, 410, 410, True, True), (64, 4, 2, 300, 300, True, True), (64, 38, 2, 1178, 1178, True, False)):
    case(*a)
