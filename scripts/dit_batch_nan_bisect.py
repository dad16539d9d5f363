#!/usr/bin/env python
"""DiT forward at batch 2: where do the NaNs come from?  usage: dit_batch_nan_bisect.py <flux|sd35>"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

torch.cuda.init()
import sdcpp_amd as sd
from bench import LEGS

leg = sys.argv[1]
mattr, wattr, lat0, ntok, cdim, ydim, ch, B, k, cfg_steps, cfg, nfwd = LEGS[leg]
if len(sys.argv) > 3:
    mattr = sys.argv[3]
sd.load_mi355x_backend()
rng = np.random.default_rng(99)


def run(lat, n, opts):
    for kk, v in opts:
        sd.backend_set_option(kk, v)
    eng = sd.Engine(model=getattr(sd, mattr), backend="MI355X0", wtype=getattr(sd, wattr), flash_attn=True)
    x = rng.standard_normal((n, ch, lat, lat)).astype(np.float32)
    t = np.full((n,), 500.0, dtype=np.float32)
    ctx = rng.standard_normal((n, ntok, cdim)).astype(np.float32)
    y = rng.standard_normal((n, ydim)).astype(np.float32)
    out = eng.unet_forward(x, t, ctx, y)
    fin = np.isfinite(out)
    per = [bool(np.isfinite(out[i]).all()) for i in range(n)]
    print(f"{leg} latent {lat} batch {n} {opts}: finite per image {per}", flush=True)
    del eng
    for kk, v in opts:
        sd.backend_set_option(kk, {"qinloop_min_rows": 513, "jit_qimages": 4096, "flash_vtr": 255, "flash_vpf": 255}.get(kk, 1))
    return fin.all()


lat = int(sys.argv[2]) if len(sys.argv) > 2 else 64
run(lat, 2, [])
for combo in ([("fuse_joint_qkv", 0), ("fuse_concat_heads", 0)], [("fuse_joint_qkv", 0), ("fuse_concat_heads", 0), ("fuse_q16", 0)], [("fuse_joint_qkv", 0), ("fuse_q16", 0)],
              [("fuse_concat_heads", 0), ("fuse_q16", 0)], [("fuse_joint_qkv", 0), ("fuse_concat_heads", 0), ("fuse_q16", 0), ("fuse_rope", 0)], [("flash_grid", 0)],
              [("flash_nsel", 0)], [("flash_vtr", 0)], [("flash_vpf", 0)]):
    run(lat, 2, combo)
