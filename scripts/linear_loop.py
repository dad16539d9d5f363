#!/usr/bin/env python
"""One Linear launched back to back for ~25 s (for clock / power sampling): usage linear_loop.py rows K M [f16|q4] [idle]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F16, F32, Q4_0, Graph
sd.load_mi355x_backend()
L = sd.lib()
rng = np.random.default_rng(0)
rows, K, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wt = Q4_0 if len(sys.argv) > 4 and sys.argv[4] == "q4" else F16
x = rng.standard_normal((rows, K)).astype(np.float32)
w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
with Graph("MI355X0") as g:
    y = L.ggml_mul_mat(g.ctx, g.weight(w, wt), g.input(x))
    g.run(y)
    gf = L.ggml_new_graph_custom(g.ctx, 256, False)
    L.ggml_build_forward_expand(gf, y)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 25:
        for _ in range(200):
            L.ggml_backend_graph_compute(g.backend, gf)
        g.fetch(y); n += 200
    dt = time.perf_counter() - t0
print(f"{n} launches in {dt:.1f} s: {dt / n * 1e6:.1f} us each, {2.0 * rows * K * M * n / dt / 1e12:.0f} TFLOP/s sustained")
