"""One shape for the rocprofv3 --pmc passes on the ping-pong flash kernel: SD1.5 64x64-level self-attention (L = 4096, d = 40, 128 (head, image) pairs)."""
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
for kv in sys.argv[1:]:
    sd.backend_set_option(kv.split("=")[0], int(kv.split("=")[1]))
rng = np.random.default_rng(0)
d, Lq, HN = 40, 4096, 128
q = rng.standard_normal((HN, Lq, d)).astype(np.float32)
k = rng.standard_normal((HN, Lq, d)).astype(np.float32)
v = rng.standard_normal((HN, Lq, d)).astype(np.float32)
with Graph("MI355X0") as g:
    node = L.ggml_flash_attn_ext(g.ctx, g.input(q), g.input(k, F16), g.input(v, F16), None, 1.0 / np.sqrt(d), 0.0, 0.0)
    g.run(node)
    gf = L.ggml_new_graph_custom(g.ctx, 64, False)
    L.ggml_build_forward_expand(gf, node)
    for _ in range(3):
        L.ggml_backend_graph_compute(g.backend, gf)
    L.ggml_backend_synchronize(g.backend)
print("done")
