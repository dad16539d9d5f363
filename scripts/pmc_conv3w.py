"""One conv for the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: 3x3, 320 -> 320 channels on 64x64 maps, 16 images, bias only (no residual, no
embedding add): the dominant launch of the SD1.5 bench step in isolation.  Options as key=int arguments (conv3w=0: the round-2 per-tap kernel)."""
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
for kv in sys.argv[1:]:
    sd.backend_set_option(kv.split("=")[0], int(kv.split("=")[1]))
rng = np.random.default_rng(0)
N, IC, OC, HW = 16, 320, 320, 64
x = rng.standard_normal((N, IC, HW, HW)).astype(np.float32)
w = (rng.standard_normal((OC, IC, 3, 3)) / np.sqrt(IC * 9)).astype(np.float32)
b = rng.standard_normal(OC).astype(np.float32)
with Graph("MI355X0") as g:
    y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 1, 1, 1, 1)
    node = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))
    g.run(node)
    gf = L.ggml_new_graph_custom(g.ctx, 64, False)
    L.ggml_build_forward_expand(gf, node)
    for _ in range(4):
        L.ggml_backend_graph_compute(g.backend, gf)
    L.ggml_backend_synchronize(g.backend)
print("done")
