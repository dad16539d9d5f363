#!/usr/bin/env python
"""Random-weight bench legs: are the model outputs finite?  One forward (unet_forward) at batch 1 / 2 / 4 and a two-step trajectory.  usage: leg_finite_check.py <leg> [key=int ...]"""
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
mattr, wattr, lat, ntok, cdim, ydim, ch, B, k, cfg_steps, cfg, nfwd = LEGS[leg]
dit = not leg.startswith("sdxl")
sd.load_mi355x_backend()
for a in sys.argv[2:]:
    sd.backend_set_option(a.split("=")[0], int(a.split("=")[1]))
eng = sd.Engine(model=getattr(sd, mattr), backend="MI355X0", wtype=getattr(sd, wattr), flash_attn=True)
rng = np.random.default_rng(99)


def stat(tag, out):
    fin = np.isfinite(out)
    print(f"{leg} {tag}: finite {bool(fin.all())} ({int((~fin).sum())} of {out.size} not), |max| of the finite {float(np.abs(out[fin]).max()) if fin.any() else float('nan'):.3e}", flush=True)


for n in (1, 2, 4):
    x = rng.standard_normal((n, ch, lat, lat)).astype(np.float32)
    t = np.full((n,), 500.0, dtype=np.float32)
    ctx = rng.standard_normal((n, ntok, cdim)).astype(np.float32)
    y = rng.standard_normal((n, ydim)).astype(np.float32)
    for scale in (1.0, 0.1):
        stat(f"forward batch {n}, context x{scale}", eng.unet_forward(x, t, ctx * scale, y * scale))
cond = rng.standard_normal((1, ntok, cdim)).astype(np.float32)
uncond = rng.standard_normal((1, ntok, cdim)).astype(np.float32)
y = rng.standard_normal((1, ydim)).astype(np.float32)
kw = dict(width=lat * 8, height=lat * 8, cfg=cfg, seed=42, batch=B, device_batch=B, method=sd.EULER if dit else sd.EULER_A, cond_y=y, uncond_y=y, fuse_cfg=True, device_sampler=True)
unc = None if nfwd == 1 else uncond
for steps in (1, 2):
    stat(f"trajectory of {steps} step(s)", eng.sample_latents(cond, unc, steps=steps, **kw))
