#!/usr/bin/env python
"""Correctness + timing of the pipelined 256x320 tile (gemm16_tile = 5) against the oracle-validated 128x128 tile (0) and the per-shape
choice without it, on the SD1.5 batch-16 shapes it is meant for.  Timing = HIP events around each dispatch (kernel_timing families)."""
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


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


MODES = ((0, 1), (-1, 0), (-1, 1))   # (gemm16_tile, gemm16_t320): the 128x128 tile; per-shape choice without / with the pipelined tiles


def run_case(label, build, flops, tiles=MODES, fams=0b100000000000111):
    outs = {}
    for tile in tiles:
        sd.backend_set_option("gemm16_tile", tile[0])
        sd.backend_set_option("gemm16_t320", tile[1])
        with Graph("MI355X0") as g:
            node = build(g)
            out = g.run(node)                       # builds weight image + plan, first result
            gf = L.ggml_new_graph_custom(g.ctx, 256, False)
            L.ggml_build_forward_expand(gf, node)
            sd.kernel_timing_enable(fams)           # conv 256 / conv 128 / linear (+ split-K reduce) families
            for _ in range(REPS):
                L.ggml_backend_graph_compute(g.backend, gf)
            t = sd.kernel_timings()
            sd.kernel_timing_enable(0)
            out2 = g.fetch(node)
        ms = sum(f["total_ms"] for f in t) / REPS
        outs[tile] = (out, out2, ms)
    base = outs[tiles[0]][0]
    line = f"{label:58s}"
    for tile in tiles:
        o, o2, ms = outs[tile]
        line += f" | {tile[0]:2d}/{tile[1]}: {ms*1e3:7.1f} us {flops/ms/1e9:7.1f} TF rel {rel_l2(o, base):.1e} rerun {'=' if np.array_equal(o, o2) else 'DIFF'}"
    print(line, flush=True)
    sd.backend_set_option("gemm16_tile", -1)
    sd.backend_set_option("gemm16_t320", 1)
    return all(rel_l2(outs[t][0], base) < 3e-5 and np.array_equal(outs[t][0], outs[t][1]) and np.isfinite(outs[t][0]).all() for t in tiles)


def conv(N, IC, OC, HW, ks=3, stride=1, res=False, ups=False, tiles=MODES):
    x = rng.standard_normal((N, IC, HW, HW)).astype(np.float32)
    w = (rng.standard_normal((OC, IC, ks, ks)) / np.sqrt(IC * ks * ks)).astype(np.float32)
    b = rng.standard_normal(OC).astype(np.float32)
    o = (HW * (2 if ups else 1)) // stride
    r = rng.standard_normal((N, OC, o, o)).astype(np.float32)

    def build(g):
        xi = g.input(x)
        if ups:
            xi = L.ggml_upscale(g.ctx, xi, 2, 0)
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), xi, stride, stride, ks // 2, ks // 2, 1, 1)
        y = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))
        if res:
            y = L.ggml_add(g.ctx, y, g.input(r))
        return y

    return run_case(f"conv{ks}x{ks} N{N} {IC}->{OC} @{HW} s{stride}{' +res' if res else ''}{' ups' if ups else ''}", build, 2.0 * N * o * o * OC * IC * ks * ks, tiles=tiles)


def linear(tokens, K, M, res=False, tiles=MODES):
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((tokens, M)).astype(np.float32)

    def build(g):
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x)), g.weight(b, F32))
        if res:
            y = L.ggml_add(g.ctx, y, g.input(r))
        return y

    return run_case(f"linear {tokens}x{K}->{M}{' +res' if res else ''}", build, 2.0 * tokens * K * M, tiles=tiles)


def geglu_ff(tokens, dim, inner):
    x = rng.standard_normal((1, tokens, dim)).astype(np.float32)
    w1 = (rng.standard_normal((2 * inner, dim)) / np.sqrt(dim)).astype(np.float32)
    b1 = rng.standard_normal(2 * inner).astype(np.float32)
    w2 = (rng.standard_normal((dim, inner)) / np.sqrt(inner)).astype(np.float32)
    b2 = rng.standard_normal(dim).astype(np.float32)

    def build(g):
        from ggml_graph import tensor_struct
        xin = g.input(x)
        h = L.ggml_mul_mat(g.ctx, g.weight(w1, F16), xin)
        h = L.ggml_add_inplace(g.ctx, h, g.weight(b1, F32))
        ts = [int(tensor_struct(h).nb[i]) for i in range(4)]
        lo = L.ggml_view_4d(g.ctx, h, inner, tokens, 1, 1, ts[1], ts[2], ts[3], 0)
        hi = L.ggml_view_4d(g.ctx, h, inner, tokens, 1, 1, ts[1], ts[2], ts[3], inner * 4)
        gate = L.ggml_gelu_inplace(g.ctx, L.ggml_cont(g.ctx, hi))
        h = L.ggml_mul(g.ctx, lo, gate)
        y = L.ggml_mul_mat(g.ctx, g.weight(w2, F16), h)
        y = L.ggml_add_inplace(g.ctx, y, g.weight(b2, F32))
        return L.ggml_add(g.ctx, y, xin)

    return run_case(f"GEGLU FF {tokens} x {dim} -> 2x{inner} -> {dim}", build, 2.0 * tokens * dim * inner * 3)


def main():
    ok = True
    # small / ragged shapes first (short K: prologue + drain paths; ragged rows; K not a multiple of 128)
    ok &= linear(300, 64, 320)
    ok &= linear(256, 96, 320)
    ok &= linear(1000, 160, 640, res=True)
    ok &= linear(77, 768, 320)
    ok &= conv(1, 64, 320, 16)
    ok &= conv(2, 320, 320, 16, res=True)
    ok &= conv(1, 32, 320, 24, ks=1)
    ok &= conv(1, 320, 320, 8, ups=True)
    # the SD1.5 batch-16 shapes of the 64x64 level
    ok &= conv(16, 320, 320, 64)
    ok &= conv(16, 320, 320, 64, res=True)
    ok &= conv(16, 640, 320, 64)
    ok &= conv(16, 960, 320, 64)
    ok &= conv(16, 320, 320, 64, ks=1, res=True)
    ok &= conv(16, 4, 320, 64)
    ok &= conv(16, 320, 320, 32, ups=True)
    ok &= linear(65536, 320, 320)
    ok &= linear(65536, 320, 320, res=True)
    ok &= linear(65536, 1280, 320, res=True)
    ok &= conv(16, 640, 640, 32)
    ok &= conv(16, 1280, 1280, 16)
    ok &= linear(16384, 640, 640)
    ok &= linear(16384, 2560, 640, res=True)
    # under-filled outputs: 256x320 tiles with K slices (32x32, 16x16 and 8x8 UNet levels)
    ok &= conv(16, 1280, 640, 32)
    ok &= conv(16, 2560, 1280, 16)
    ok &= conv(16, 1280, 1280, 16, res=True)
    ok &= conv(16, 1280, 1280, 8)
    ok &= conv(16, 2560, 1280, 8)
    ok &= conv(2, 1280, 1280, 16)
    ok &= linear(4096, 5120, 1280, res=True)
    ok &= linear(1232, 768, 1280)
    # GEGLU feed-forwards (FF1 on the pipelined 256x256 tile)
    ok &= geglu_ff(300, 320, 1280)
    ok &= geglu_ff(65536, 320, 1280)
    ok &= geglu_ff(16384, 640, 2560)
    ok &= geglu_ff(4096, 1280, 5120)
    print("ALL OK" if ok else "MISMATCH", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
