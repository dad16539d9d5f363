#!/usr/bin/env python
"""One DiT forward (for kernel traces): usage dit_one_forward.py <model attr> <wtype attr> <latent> <batch>"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

torch.cuda.init()
import sdcpp_amd as sd

mattr, wattr, lat, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
sd.load_mi355x_backend()
sd.backend_set_option("hip_graph", 0)
rng = np.random.default_rng(99)
flux = "FLUX" in mattr
ntok, cdim, ydim, ch = (256, 4096, 768, 16) if flux else (154, 4096, 2048, 16)
x = rng.standard_normal((n, ch, lat, lat)).astype(np.float32)
t = np.full((n,), 0.5 if flux else 500.0, dtype=np.float32)
ctx = rng.standard_normal((n, ntok, cdim)).astype(np.float32)
y = rng.standard_normal((n, ydim)).astype(np.float32)
eng = sd.Engine(model=getattr(sd, mattr), backend="MI355X0", wtype=getattr(sd, wattr), flash_attn=True)
out = eng.unet_forward(x, t, ctx, y)
print(f"{mattr} latent {lat} batch {n}: finite {bool(np.isfinite(out).all())}", flush=True)
st = sd.backend_stats()
print({k: st[k] for k in ("flash_slice_images", "flash_out_alias", "fused_attention", "qinloop_linears", "qgemm16_linears")}, flush=True)
