#!/usr/bin/env python
"""Quantised Linears (q8_0 / q4_0 GGUF blocks) on the three paths the planner can take, per row count, on the text-stream / modulation shapes of
FLUX and SD3.5:  raw-gemv = k_qgemv / k_qgemv_rows (<= 16 rows, VALU, one launch);  raw-mfma = k_qgemm16 (raw blocks dequantised in registers into
MFMA fragments);  image = f16 weight image + k_gemm16.  Time = HIP events around each dispatch summed over ALL kernel families of the graph (so
pack / split-K passes count), with the split per family.  This is the measurement behind the planner's defaults (qgemv_max_rows, qgemm16_max_rows)."""
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
REPS = 5
MODES = {"raw-gemv": dict(qgemv=1, qgemv_max_rows=16, qgemm16_max_rows=0), "raw-mfma": dict(qgemv=1, qgemv_max_rows=2, qgemm16_max_rows=4096),
         "image": dict(qgemv=0, qgemv_max_rows=16, qgemm16_max_rows=0)}


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def case(rows, K, M, wtype):
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    res = {}
    for mode, opts in MODES.items():
        if mode == "raw-gemv" and rows > 16:
            continue
        for k, v in opts.items():
            sd.backend_set_option(k, v)
        with Graph("MI355X0") as g:
            node = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), g.input(x)), g.weight(b, F32))
            out = g.run(node)
            gf = L.ggml_new_graph_custom(g.ctx, 64, False)
            L.ggml_build_forward_expand(gf, node)
            sd.kernel_timing_enable(sd.KF_ALL)
            for _ in range(REPS):
                L.ggml_backend_graph_compute(g.backend, gf)
            t = sd.kernel_timings()
            sd.kernel_timing_enable(0)
        split = " + ".join(f"{f['kernel'].split('(')[0].strip()[:14]} {f['total_ms'] / REPS * 1e3:.1f}" for f in t if f["total_ms"] > 0)
        res[mode] = (out, sum(f["total_ms"] for f in t) / REPS * 1e3, split)
    for k, v in MODES["raw-gemv"].items():
        sd.backend_set_option(k, v)
    wb = M * K // 32 * (34 if wtype == Q8_0 else 18)
    line = f"{'q8_0' if wtype == Q8_0 else 'q4_0'} rows={rows:4d} K={K:5d} M={M:5d} ({wb / 1e6:5.1f} MB quantised)"
    for mode, (out, us, split) in res.items():
        line += f" | {mode}: {us:6.1f} us [{split}] rel {rel_l2(out, res['image'][0]):.0e}"
    print(line, flush=True)


shapes = [(3072, 9216), (3072, 18432), (12288, 3072), (4096, 3072)]
for wtype in (Q8_0, Q4_0):
    for K, M in shapes:
        for rows in (1, 2, 4, 8, 16, 32, 77, 256):
            case(rows, K, M, wtype)
