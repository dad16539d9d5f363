#!/usr/bin/env python
"""k_fgemv (few-row f16 / f32 Linear, optional deferred SiLU) against the GEMM path it replaces (SiLU + f16 pack + k_gemm16 + split-K reduce) on the
SD1.5 / SDXL embedding shapes: HIP-event time summed over all kernel families of the graph, and the split per family."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F16, F32, Graph

sd.load_mi355x_backend()
L = sd.lib()
rng = np.random.default_rng(0)
REPS = 10


def case(rows, K, M, wtype, silu):
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    line = f"{'f16' if wtype == F16 else 'f32'} rows={rows:2d} K={K:4d} M={M:4d} silu={int(silu)}"
    outs = []
    for mode in (1, 0):
        sd.backend_set_option("fgemv", mode)
        with Graph("MI355X0") as g:
            h = g.input(x)
            if silu:
                h = L.ggml_silu(g.ctx, h)
            node = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), h), g.weight(b, F32))
            outs.append(g.run(node))
            gf = L.ggml_new_graph_custom(g.ctx, 64, False)
            L.ggml_build_forward_expand(gf, node)
            sd.kernel_timing_enable(sd.KF_ALL)
            for _ in range(REPS):
                L.ggml_backend_graph_compute(g.backend, gf)
            t = sd.kernel_timings()
            sd.kernel_timing_enable(0)
        split = " + ".join(f"{f['kernel'].split('(')[0].strip()[:12]} {f['total_ms'] / REPS * 1e3:.1f}" for f in t if f["total_ms"] > 0)
        line += f" | {'k_fgemv' if mode else 'gemm path'}: {sum(f['total_ms'] for f in t) / REPS * 1e3:6.1f} us [{split}]"
    sd.backend_set_option("fgemv", 1)
    a, b2 = (np.asarray(o, np.float64).ravel() for o in outs)
    print(line + f" | rel {np.linalg.norm(a - b2) / np.linalg.norm(b2):.1e}", flush=True)


for wtype in (F16, F32):
    for rows in (2, 4, 8, 16):
        for K, M, silu in ((1280, 1280, True), (1280, 320, True), (320, 1280, False), (1280, 640, False), (2816, 1280, False)):
            case(rows, K, M, wtype, silu)
