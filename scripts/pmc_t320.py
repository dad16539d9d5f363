#!/usr/bin/env python
"""One conv shape on the pipelined 256x320 tile, a few launches — for rocprofv3 --pmc passes (scripts/gpu_pmc_t320.sh)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from ggml_graph import F16, Graph

sd.load_mi355x_backend()
L = sd.lib()
rng = np.random.default_rng(0)
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, IC, OC, HW = 16, 960, 320, 64
x = rng.standard_normal((N, IC, HW, HW)).astype(np.float32)
w = (rng.standard_normal((OC, IC, 3, 3)) / np.sqrt(IC * 9)).astype(np.float32)
if abl:
    sd.backend_set_option("gemm16_abl", abl)
with Graph("MI355X0") as g:
    y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 1, 1, 1, 1)
    g.run(y)
    gf = L.ggml_new_graph_custom(g.ctx, 256, False)
    L.ggml_build_forward_expand(gf, y)
    for _ in range(3):
        L.ggml_backend_graph_compute(g.backend, gf)
