import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import sdcpp_amd as sd
sd.load_mi355x_backend()
sd.load_backend(ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so")
rng = np.random.default_rng(0)
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
ref_e = sd.Engine(model=sd.SD15_TINY, backend="CPU-oracle")
for hw in (8, 16, 32):
    for n in (1, 2, 3):
        x = rng.standard_normal((n, 4, hw, hw)).astype(np.float32)
        t = np.full(n, 500.0, np.float32)
        c = rng.standard_normal((1, 77, 64)).astype(np.float32)
        ref = ref_e.unet_forward(x, t, c)
        res = []
        for opt in (1, 0):
            sd.backend_set_option("fuse_chan_add", opt)
            e = sd.Engine(model=sd.SD15_TINY, backend="MI355X0")
            res.append(rel(e.unet_forward(x, t, c), ref))
        print(f"latent {hw}x{hw} N={n}: fuse_chan_add=1 rel {res[0]:.2e} | =0 rel {res[1]:.2e}", flush=True)
sd.backend_set_option("fuse_chan_add", 1)
