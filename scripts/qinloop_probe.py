#!/usr/bin/env python
"""Quantised Linears above k_qgemm16's row range, per launch (HIP events on the launch stream): the f16-image GEMM (cached image / just-in-time rebuild) against
the in-loop dequantisation of the pipelined 256 x 256 tile.  usage: qinloop_probe.py rows K M q8|q4 [nweights]
nweights different weights read the same activation in one graph (default 6: more weight bytes than the 256 MB Infinity Cache holds, as inside a model)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F32, Q4_0, Q8_0, Graph

sd.load_mi355x_backend()
L = sd.lib()
rng = np.random.default_rng(0)
rows, K, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wtype = Q8_0 if sys.argv[4] == "q8" else Q4_0
nw = int(sys.argv[5]) if len(sys.argv) > 5 else 6
x = rng.standard_normal((rows, K)).astype(np.float32)
ws = [(rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32) for _ in range(nw)]
b = rng.standard_normal(M).astype(np.float32)


def build(g):
    xin = g.input(x)
    acc = None
    for w in ws:
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), xin), g.weight(b, F32))
        acc = y if acc is None else L.ggml_add(g.ctx, acc, y)
    return acc


sets = [("image cached", [("qinloop_min_rows", 0), ("jit_qimages", 0)]), ("image rebuilt per launch", [("qinloop_min_rows", 0), ("jit_qimages", 1)]),
        ("in-loop dequantisation", [("qinloop_min_rows", 513), ("jit_qimages", 4096)])]
ref = None
for rnd in range(2):
    for name, opts in sets:
        for k, v in opts:
            sd.backend_set_option(k, v)
        with Graph("MI355X0") as g:
            node = build(g)
            out = g.run(node)
            gf = L.ggml_new_graph_custom(g.ctx, 256, False)
            L.ggml_build_forward_expand(gf, node)
            sd.kernel_timing_enable(sd.KF_ALL)
            for _ in range(5):
                L.ggml_backend_graph_compute(g.backend, gf)
            t = sd.kernel_timings()
            sd.kernel_timing_enable(0)
        if ref is None:
            ref = out
        lin = [f for f in t if "Linear MFMA" in f["kernel"]]
        pk = [f for f in t if "operand image" in f["kernel"] and "LayerNorm" not in f["kernel"]]
        lin_us = sum(f["total_ms"] for f in lin) / 5 / nw * 1e3
        pk_us = sum(f["total_ms"] for f in pk) / 5 / nw * 1e3
        tf = 2.0 * rows * K * M / (lin_us * 1e-6) / 1e12
        print(f"rows {rows} K {K} M {M} {sys.argv[4]} {name:26s}: GEMM {lin_us:8.1f} us ({tf:6.0f} TFLOP/s) + pack/rebuild {pk_us:7.1f} us per Linear; identical to first: {bool(np.array_equal(out, ref))}", flush=True)
sd.backend_set_option("qinloop_min_rows", 513)
sd.backend_set_option("jit_qimages", 4096)
