#!/usr/bin/env python
"""Per-kernel-family table of one TAESD decode (HIP events around every dispatch, eager): usage tae_probe.py [model SD15_TINY|SD35_TINY] [latent] [batch]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

torch.cuda.init()
import sdcpp_amd as sd

model = sys.argv[1] if len(sys.argv) > 1 else "SD15_TINY"
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd.load_mi355x_backend()
e = sd.Engine(model=getattr(sd, model), backend="MI355X0")
ch = 16 if model.startswith(("SD35", "FLUX")) else 4
z = np.random.default_rng(0).standard_normal((B, ch, lat, lat)).astype(np.float32)
e.tae_decode(z)
e.tae_decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    e.tae_decode(z)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
sd.kernel_timing_enable(sd.KF_ALL)
e.tae_decode(z)
fams = sorted(sd.kernel_timings(), key=lambda f: -f["total_ms"])
sd.kernel_timing_enable(0)
tot = sum(f["total_ms"] for f in fams)
# 2 * rows * K * M per conv: conv_in (zc -> 64), 10 blocks x 3 convs + 3 stage convs (64 -> 64), conv_out (64 -> 3); 3 blocks at each of the first three resolutions, 1 at the last
hw = [lat * lat * (4 ** s) for s in range(4)]
fl = 2 * 9 * (hw[0] * ch * 64 + sum(hw[s] * 64 * 64 * 9 for s in range(3)) + sum(hw[s + 1] * 64 * 64 for s in range(3)) + hw[3] * 64 * 64 * 3 + hw[3] * 64 * 3) * B
print(f"{model} TAESD decode {lat*8}x{lat*8} batch {B}: {ms:.2f} ms wall (incl. D2H of {B*3*lat*lat*64*4/1e6:.0f} MB), kernels {tot:.2f} ms, {fl/1e12:.3f} TFLOP -> {fl/(tot/1e3)/1e12:.0f} TFLOP/s over the kernels")
for f in fams:
    sec = f["total_ms"] * 1e-3
    rate = f"{f['total_flops']/sec/1e12:7.1f} TFLOP/s" if f["bound"] == "mfma" else f"{f['total_bytes']/sec/1e9:7.1f} GB/s"
    print(f"   {f['kernel'][:84]:84s} {f['launches']:4d} launches {f['total_ms']:8.3f} ms  {rate}")
st = sd.backend_stats()
print("   backend:", {k: st[k] for k in ("kernels_planned", "nodes_seen", "fused_conv_bounced", "window_convs", "split_k_gemms") if k in st})
