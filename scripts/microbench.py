#!/usr/bin/env python
"""Micro-benchmark of the hot kernels at SD1.5 (batch 16 = 8 images x cond/uncond) shapes through the C ABI:
each case is a 1-op ggml graph (conv chain / linear / flash-attn) computed REPS times; wall time per compute (sync) is printed
together with the algorithmic TFLOP/s.  Run under rocprofv3 for exact kernel times."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F16, F32, Graph

sd.load_mi355x_backend()
L = sd.lib()
blib = C.CDLL(str(sd.BACKEND_LIB))
blib.ggml_backend_mi355x_set_option.argtypes = [C.c_char_p, C.c_int]
REPS = 10
rng = np.random.default_rng(0)


def timed(g, node, flops, label):
    gf = L.ggml_new_graph_custom(g.ctx, 256, False)
    L.ggml_set_output(node)
    L.ggml_build_forward_expand(gf, node)
    if g._weights:
        g._wbuf = L.ggml_backend_alloc_ctx_tensors(g.wctx, g.backend)
        L.ggml_backend_buffer_set_usage(g._wbuf, 1)
        for t, raw in g._weights:
            L.ggml_backend_tensor_set(t, raw, 0, len(raw))
    g._galloc = L.ggml_gallocr_new(L.ggml_backend_get_default_buffer_type(g.backend))
    assert L.ggml_gallocr_alloc_graph(g._galloc, gf)
    for t, raw in g._inputs:
        L.ggml_backend_tensor_set(t, raw, 0, len(raw))
    for _ in range(3):
        L.ggml_backend_graph_compute(g.backend, gf)
    t0 = time.perf_counter()
    for _ in range(REPS):
        L.ggml_backend_graph_compute(g.backend, gf)
    dt = (time.perf_counter() - t0) / REPS
    print(f"{label:52s} {dt*1e6:9.1f} us  {flops/dt/1e12:8.1f} TFLOP/s (wall, incl. pack+launch+sync)", flush=True)


def conv(N, IC, OC, HW, ks=3, stride=1):
    x = rng.standard_normal((N, IC, HW, HW)).astype(np.float32)
    w = (rng.standard_normal((OC, IC, ks, ks)) / np.sqrt(IC * ks * ks)).astype(np.float32)
    b = np.zeros(OC, np.float32)
    with Graph("MI355X0") as g:
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), stride, stride, ks // 2, ks // 2, 1, 1)
        y = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))
        o = HW // stride
        timed(g, y, 2.0 * N * o * o * OC * IC * ks * ks, f"conv{ks}x{ks} N{N} {IC}->{OC} @{HW}x{HW} s{stride}")


def linear(tokens, K, M):
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = np.zeros(M, np.float32)
    with Graph("MI355X0") as g:
        y = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))
        y = L.ggml_add_inplace(g.ctx, y, g.weight(b, F32))
        timed(g, y, 2.0 * tokens * K * M, f"linear tok{tokens} {K}->{M}")


def flash(d, L_, HN):
    q = rng.standard_normal((HN, L_, d)).astype(np.float32)
    k = rng.standard_normal((HN, L_, d)).astype(np.float32)
    v = rng.standard_normal((HN, L_, d)).astype(np.float32)
    with Graph("MI355X0") as g:
        y = L.ggml_flash_attn_ext(g.ctx, g.input(q), g.input(k, F16), g.input(v, F16), None, 1.0 / np.sqrt(d), 0.0, 0.0)
        timed(g, y, 4.0 * L_ * L_ * d * HN, f"flash d{d} L{L_} HN{HN}")


if __name__ == "__main__":
    variants = [int(v) for v in sys.argv[1:]] or [1]
    for v in variants:
        blib.ggml_backend_mi355x_set_option(b"gemm16_variant", v)
        print(f"--- gemm16 variant {v}")
        conv(16, 320, 320, 64)
        conv(16, 640, 640, 32)
        conv(16, 1280, 1280, 16)
        conv(16, 1280, 1280, 8)
        conv(16, 960, 320, 64)
        conv(16, 320, 320, 64, ks=1)
        linear(65536, 320, 320)
        linear(65536, 320, 2560)
        linear(65536, 1280, 320)
        linear(16384, 640, 5120)
        linear(4096, 1280, 10240)
        linear(1232, 768, 320)
    flash(40, 4096, 128)
    flash(80, 1024, 128)
    flash(160, 256, 128)
