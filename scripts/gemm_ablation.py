#!/usr/bin/env python
"""Which pipeline bounds the pipelined 256x320 tile?  Needs the experiment build of the backend:
    SDCPP_BUILD_VARIANT=exp python -c "from sdcpp_amd import build; build.build_all()"
    SDCPP_BACKEND_LIB=stable-diffusion.cpp_amd/lib_exp/libggml-mi355x.so python scripts/gemm_ablation.py
Times the SD1.5 64x64-level convs three ways (HIP events per dispatch): product kernel, DMA + reads + barriers without MFMAs (option
gemm16_abl = 1), reads + MFMAs without DMA after the pipeline fill (2), input tiles
fetched for one tap of nine (3: the DMA volume of a kernel that keeps the input window resident in LDS).  The ablations compute WRONG results on purpose."""
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


def conv_time(N, IC, OC, HW, abl):
    x = rng.standard_normal((N, IC, HW, HW)).astype(np.float32)
    w = (rng.standard_normal((OC, IC, 3, 3)) / np.sqrt(IC * 9)).astype(np.float32)
    sd.backend_set_option("gemm16_abl", abl)
    with Graph("MI355X0") as g:
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 1, 1, 1, 1)
        g.run(y)
        gf = L.ggml_new_graph_custom(g.ctx, 256, False)
        L.ggml_build_forward_expand(gf, y)
        sd.kernel_timing_enable(0b1)
        for _ in range(5):
            L.ggml_backend_graph_compute(g.backend, gf)
        t = sd.kernel_timings()
        sd.kernel_timing_enable(0)
    sd.backend_set_option("gemm16_abl", 0)
    return sum(f["total_ms"] for f in t) / 5 * 1e3


def linear_time(rows, K, M, abl):
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    sd.backend_set_option("gemm16_abl", abl)
    sd.backend_set_option("gemm16_t192p", 0)
    with Graph("MI355X0") as g:
        y = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))
        g.run(y)
        gf = L.ggml_new_graph_custom(g.ctx, 256, False)
        L.ggml_build_forward_expand(gf, y)
        sd.kernel_timing_enable(0b100)
        for _ in range(5):
            L.ggml_backend_graph_compute(g.backend, gf)
        t = sd.kernel_timings()
        sd.kernel_timing_enable(0)
    sd.backend_set_option("gemm16_abl", 0)
    sd.backend_set_option("gemm16_t192p", 1)
    return sum(f["total_ms"] for f in t) / 5 * 1e3


if len(sys.argv) > 1 and sys.argv[1] == "linear":
    # the pipelined 256 x 256 Linear tile (DiT shapes): whole rounds (4096 x 3072 -> 12288: 768 tiles = 3 rounds) and a partial one
    for rows, K, M in ((4096, 3072, 12288), (4096, 12288, 3072), (8192, 2432, 9728)):
        full, nomfma, nodma, nobar, nowait = (linear_time(rows, K, M, a) for a in (0, 1, 2, 5, 6))
        fl = 2.0 * rows * K * M
        print(f"linear {rows}x{K}->{M}: product {full:7.1f} us ({fl / full / 1e6:6.0f} TFLOP/s) | DMA + fragment reads + barriers, no MFMA {nomfma:7.1f} us | reads + MFMAs, no DMA after the fill {nodma:7.1f} us ({fl / nodma / 1e6:6.0f} TFLOP/s) | ... and no barrier {nobar:7.1f} us ({fl / nobar / 1e6:6.0f}) | ... and no LDS waits {nowait:7.1f} us ({fl / nowait / 1e6:6.0f})", flush=True)
    sys.exit(0)

for (N, IC, OC, HW) in ((16, 320, 320, 64), (16, 960, 320, 64), (16, 640, 640, 32)):
    full, nomfma, nodma, areuse = (conv_time(N, IC, OC, HW, a) for a in (0, 1, 2, 3))
    stages = IC // 32 * 9 if IC % 64 == 0 else (IC + 63) // 64 * 2 * 9
    print(f"conv3x3 N{N} {IC}->{OC} @{HW}: product {full:7.1f} us | DMA + reads + barriers, no MFMA {nomfma:7.1f} us | reads + MFMA, no DMA {nodma:7.1f} us "
          f"| input tile fetched for tap 0 only (A traffic / 9) {areuse:7.1f} us "
          f"| per stage ({stages} stages): {full / stages * 1e3:.0f} / {nomfma / stages * 1e3:.0f} / {nodma / stages * 1e3:.0f} / {areuse / stages * 1e3:.0f} ns", flush=True)

