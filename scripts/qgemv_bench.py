#!/usr/bin/env python
"""Achieved HBM bandwidth of the in-register dequant kernel (k_qgemv) on the FLUX / SDXL few-row Linear shapes: algorithmic bytes =
raw quantised weight bytes + activation + output, time = HIP events around each dispatch."""
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
for name, wt, rows, K, M in (("FLUX double-block modulation q4_0", Q4_0, 1, 3072, 18432), ("FLUX single-block modulation q4_0", Q4_0, 1, 3072, 9216),
                             ("SD3.5 adaLN pair q8_0", Q8_0, 2, 2432 // 256 * 256, 14592), ("SDXL ResBlock emb q8_0", Q8_0, 2, 1280, 1280), ("q8_0 3072 -> 18432", Q8_0, 1, 3072, 18432)):
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    with Graph("MI355X0") as g:
        y = L.ggml_mul_mat(g.ctx, g.weight(w, wt), g.input(x))
        g.run(y)
        gf = L.ggml_new_graph_custom(g.ctx, 64, False)
        L.ggml_build_forward_expand(gf, y)
        sd.kernel_timing_enable(1 << 4)
        for _ in range(20):
            L.ggml_backend_graph_compute(g.backend, gf)
        t = sd.kernel_timings()
        sd.kernel_timing_enable(0)
    if t:
        f = t[0]
        print(f"{name:40s} rows {rows} K {K:5d} M {M:6d}: {f['total_ms'] / f['launches'] * 1e3:7.1f} us  {f['total_bytes'] / (f['total_ms'] * 1e-3) / 1e9:7.1f} GB/s", flush=True)
    else:
        print(name, "not on the qgemv kernel")
