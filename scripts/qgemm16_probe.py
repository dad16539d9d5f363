#!/usr/bin/env python
"""k_qgemm16 (raw q8_0 / q4_0 blocks, in-register dequant on the way into the MFMA units) against the f16-weight-image GEMM on the text-stream
Linear shapes of FLUX / SD3.5 / T5: time per launch (HIP events around each dispatch, all families summed so the image path's pack + split-K
passes count) and the weight-stream rate.  Finds the row count where the image path takes over (option qgemm16_max_rows)."""
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


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def case(rows, K, M, wtype):
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    res = {}
    for mode in (1, 0):
        sd.backend_set_option("qgemm16", mode)
        sd.backend_set_option("qgemm16_max_rows", 4096)
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
        res[mode] = (out, sum(f["total_ms"] for f in t) / REPS * 1e3)
    sd.backend_set_option("qgemm16", 1)
    sd.backend_set_option("qgemm16_max_rows", 512)
    wb = M * K // 32 * (34 if wtype == Q8_0 else 18)
    q_us, i_us = res[1][1], res[0][1]
    print(f"{'q8_0' if wtype == Q8_0 else 'q4_0'} rows={rows:5d} K={K:5d} M={M:5d} | raw blocks {q_us:7.1f} us ({wb / q_us / 1e3:7.1f} GB/s of quantised weights, "
          f"{2.0 * rows * K * M / q_us / 1e6:6.1f} TF) | f16 image {i_us:7.1f} us ({2.0 * M * K / i_us / 1e3:7.1f} GB/s of image) | rel {rel_l2(res[1][0], res[0][0]):.1e}", flush=True)


shapes = [(3072, 9216), (3072, 12288), (12288, 3072), (4096, 3072)]
for wtype in (Q8_0, Q4_0):
    for K, M in shapes:
        for rows in (4, 32, 77, 128, 256, 512, 1024, 2048):
            case(rows, K, M, wtype)
