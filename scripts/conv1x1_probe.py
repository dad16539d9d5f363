#!/usr/bin/env python
"""1x1 conv (SpatialTransformer proj_out: NHWC f16 operand image -> NCHW f32 + bias + residual) against a Linear of the same shape and bytes
(row-major f32 + bias + residual), per tile configuration (option gemm16_tile).  Timing: HIP events around each dispatch."""
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
REPS = 5
TILES = {-1: "auto", 0: "T128", 1: "T256", 3: "T160", 4: "T160N", 5: "T320"}


def run(build, fams, flops):
    out = {}
    for tile, name in TILES.items():
        sd.backend_set_option("gemm16_tile", tile)
        with Graph("MI355X0") as g:
            node = build(g)
            res = g.run(node)
            gf = L.ggml_new_graph_custom(g.ctx, 256, False)
            L.ggml_build_forward_expand(gf, node)
            sd.kernel_timing_enable(fams)
            for _ in range(REPS):
                L.ggml_backend_graph_compute(g.backend, gf)
            t = sd.kernel_timings()
            sd.kernel_timing_enable(0)
        ms = sum(f["total_ms"] for f in t) / REPS
        out[name] = (ms * 1e3, res)
    sd.backend_set_option("gemm16_tile", -1)
    return out


def case(N, C, HW):
    x = rng.standard_normal((N, C, HW, HW)).astype(np.float32)
    w = (rng.standard_normal((C, C, 1, 1)) / np.sqrt(C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    r = rng.standard_normal((N, C, HW, HW)).astype(np.float32)
    flops = 2.0 * N * HW * HW * C * C

    def conv(g):
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 0, 0, 1, 1)
        y = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, C, 1))
        return L.ggml_add(g.ctx, y, g.input(r))

    xt = rng.standard_normal((N, HW * HW, C)).astype(np.float32)
    rt = rng.standard_normal((N, HW * HW, C)).astype(np.float32)

    def lin(g):
        y = L.ggml_mul_mat(g.ctx, g.weight(w.reshape(C, C), F16), g.input(xt))
        y = L.ggml_add_inplace(g.ctx, y, g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(rt))

    tc = run(conv, (1 << 0) | (1 << 1) | (1 << 14), flops)
    tl = run(lin, (1 << 2) | (1 << 14), flops)
    ref = tc["auto"][1]
    print(f"N{N} C{C} {HW}x{HW}: 1x1 conv " + "  ".join(f"{k} {v[0]:6.1f}us" for k, v in tc.items()) + "  | Linear " + "  ".join(f"{k} {v[0]:6.1f}us" for k, v in tl.items())
          + f"  | conv variants max diff {max(float(np.abs(v[1] - ref).max()) for v in tc.values()):.1e}", flush=True)


if __name__ == "__main__":
    print({i: f for i, f in enumerate(sd.KERNEL_FAMILIES)} if hasattr(sd, "KERNEL_FAMILIES") else "")
    case(16, 320, 64)
    case(16, 640, 32)
    case(16, 1280, 16)
    case(2, 640, 64)
    case(2, 1280, 32)
