#!/usr/bin/env python
"""Per-kernel-family time (HIP events around every dispatch) of one hot-path call: `sd15` = the bench forward; `vae` = KL-VAE decode 64x64 -> 512x512 of 8 images,
`sdxl` = SDXL UNet 1024x1024 cond+uncond pair (q8_0), `sd35` / `flux` = one DiT forward.  usage: family_times.py vae [sdxl ...]
MI355X_KTIME_DUMP=<file> additionally appends one line per distinct launch shape (by flops / bytes)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import sdcpp_amd as sd

sd.load_mi355x_backend()
rng = np.random.default_rng(0)


def report(label, fn, reps=2):
    fn()
    sd.kernel_timing_enable(sd.KF_ALL)
    for _ in range(reps):
        fn()
    fams = sd.kernel_timings()
    sd.kernel_timing_enable(0)
    tot = sum(f["total_ms"] for f in fams)
    print(f"== {label}: {tot / reps:.2f} ms of kernels per call")
    for f in sorted(fams, key=lambda f: -f["total_ms"]):
        sec = f["total_ms"] * 1e-3
        rate = f"{f['total_flops'] / sec / 1e12:7.1f} TFLOP/s" if f["bound"] == "mfma" else f"{f['total_bytes'] / sec / 1e9:7.1f} GB/s"
        print(f"   {f['kernel'][:62]:62s} {f['launches'] / reps:6.1f} launches {f['total_ms'] / reps:8.3f} ms {100 * f['total_ms'] / tot:5.1f} %  {rate}", flush=True)


args = [a for a in sys.argv[1:] if "=" not in a]
for kv in (a for a in sys.argv[1:] if "=" in a):  # backend options: key=int
    sd.backend_set_option(kv.split("=")[0], int(kv.split("=")[1]))
for what in args or ["vae"]:
    if what == "vae":
        e = sd.Engine(model=sd.SD15, flash_attn=True)
        z = rng.standard_normal((8, 4, 64, 64)).astype(np.float32) * 0.5
        report("KL-VAE decode 64x64 -> 512x512, 8 images", lambda: e.vae_decode(z))
    elif what == "sd15":  # the bench workload's forward: 8 images x (cond, uncond) at 512x512
        e = sd.Engine(model=sd.SD15, flash_attn=True)
        x = rng.standard_normal((16, 4, 64, 64)).astype(np.float32)
        t = np.full(16, 500.0, np.float32)
        c = rng.standard_normal((16, 77, 768)).astype(np.float32)
        report("SD1.5 UNet 512x512, 8 images x cond+uncond", lambda: e.unet_forward(x, t, c, None))
    elif what == "sdxl":
        e = sd.Engine(model=sd.SDXL, wtype=sd.Q8_0, flash_attn=True)
        x = rng.standard_normal((2, 4, 128, 128)).astype(np.float32)
        t = np.full(2, 500.0, np.float32)
        c = rng.standard_normal((2, 77, 2048)).astype(np.float32)
        y = rng.standard_normal((2, 2816)).astype(np.float32)
        report("SDXL UNet 1024x1024, cond+uncond pair, q8_0", lambda: e.unet_forward(x, t, c, y))
    elif what in ("sd35", "flux"):
        flux = what == "flux"
        e = sd.Engine(model=sd.FLUX_DEV if flux else sd.SD35_LARGE, wtype=sd.Q4_0 if flux else sd.BF16, flash_attn=True)
        n = 1 if flux else 2
        x = rng.standard_normal((n, 16, 128, 128)).astype(np.float32)
        t = np.full(n, 0.5 if flux else 500.0, np.float32)
        c = rng.standard_normal((n if not flux else 1, 256 if flux else 154, 4096)).astype(np.float32)
        y = rng.standard_normal((n if not flux else 1, 768 if flux else 2048)).astype(np.float32)
        report(f"{what} forward 1024x1024 ({'1 image' if flux else 'cond+uncond pair'})", lambda: e.unet_forward(x, t, c, y))
    print({k: v for k, v in sd.backend_stats().items() if k in ("swizzled_weight_bytes", "qgemv_linears", "generic_matmul", "fused_attention")})
