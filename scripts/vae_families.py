#!/usr/bin/env python
"""Per-kernel-family table of one KL-VAE decode (HIP events around every dispatch, eager): usage vae_families.py [model SD15|SDXL|SD35_WIDE2] [latent] [batch]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

torch.cuda.init()
import sdcpp_amd as sd

model = sys.argv[1] if len(sys.argv) > 1 else "SDXL"
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd.load_mi355x_backend()
e = sd.Engine(model=getattr(sd, model), backend="MI355X0", wtype=sd.Q8_0 if model == "SDXL" else sd.F16, flash_attn=True)
ch = 16 if model.startswith(("SD35", "FLUX")) else 4
z = np.random.default_rng(0).standard_normal((B, ch, lat, lat)).astype(np.float32)
e.vae_decode(z)
e.vae_decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    e.vae_decode(z)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
sd.kernel_timing_enable(sd.KF_ALL)
e.vae_decode(z)
fams = sorted(sd.kernel_timings(), key=lambda f: -f["total_ms"])
sd.kernel_timing_enable(0)
tot = sum(f["total_ms"] for f in fams)
tflop = {64: 2.515, 128: 10.47}.get(lat, 0) * B
print(f"{model} VAE decode {lat*8}x{lat*8} batch {B}: {ms:.2f} ms wall (incl. D2H of {B*3*lat*lat*64*4/1e6:.0f} MB), kernels {tot:.2f} ms, {tflop/(ms/1e3):.0f} TFLOP/s = {tflop/(ms/1e3)/2500:.3f} of the MFMA peak")
for f in fams:
    sec = f["total_ms"] * 1e-3
    rate = f"{f['total_flops']/sec/1e12:7.1f} TFLOP/s" if f["bound"] == "mfma" else f"{f['total_bytes']/sec/1e9:7.1f} GB/s"
    print(f"   {f['kernel'][:84]:84s} {f['launches']:4d} launches {f['total_ms']:8.3f} ms  {rate}")
