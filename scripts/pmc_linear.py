"""One Linear for rocprofv3 --pmc passes: tokens x K -> M, f16 weights, bias (+ residual).  usage: pmc_linear.py tokens K M [res] [key=int ...]"""
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
pos = [a for a in sys.argv[1:] if "=" not in a]
for kv in (a for a in sys.argv[1:] if "=" in a):
    sd.backend_set_option(kv.split("=")[0], int(kv.split("=")[1]))
tokens, K, M = int(pos[0]), int(pos[1]), int(pos[2])
res = len(pos) > 3 and pos[3] == "res"
rng = np.random.default_rng(0)
x = rng.standard_normal((1, tokens, K)).astype(np.float32)
w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
b = rng.standard_normal(M).astype(np.float32)
r = rng.standard_normal((1, tokens, M)).astype(np.float32)
with Graph("MI355X0") as g:
    y = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))
    node = L.ggml_add_inplace(g.ctx, y, g.weight(b, F32))
    if res:
        node = L.ggml_add(g.ctx, node, g.input(r))
    g.run(node)
    gf = L.ggml_new_graph_custom(g.ctx, 64, False)
    L.ggml_build_forward_expand(gf, node)
    for _ in range(4):
        L.ggml_backend_graph_compute(g.backend, gf)
    L.ggml_backend_synchronize(g.backend)
print("done")
