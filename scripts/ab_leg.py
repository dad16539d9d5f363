#!/usr/bin/env python
"""Within-process interleaved A/B of a backend option on one of bench.py's other workloads (device-resident sampler steps, hipGraph replay like the product).
usage: ab_leg.py <leg: sdxl|sdxl_b8|flux|sd35> <option> <v0,v1,...> [rounds] [steps]     e.g.  ab_leg.py flux hoist_mod 0,1 3 5
Prints ms per step per setting, the few-row / dominant family times (HIP events, eager pass), and the max |difference| of the sampled latents between settings."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

torch.cuda.init()
import sdcpp_amd as sd
from bench import LEGS

leg, key = sys.argv[1], sys.argv[2]
vals = [int(v) for v in sys.argv[3].split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
mattr, wattr, lat, ntok, cdim, ydim, ch, B, k, cfg_steps, cfg, nfwd = LEGS[leg]
dit = not leg.startswith("sdxl")
sd.load_mi355x_backend()
eng = sd.Engine(model=getattr(sd, mattr), backend="MI355X0", wtype=getattr(sd, wattr), flash_attn=True)
rng = np.random.default_rng(99)
cond = rng.standard_normal((1, ntok, cdim)).astype(np.float32)
uncond = rng.standard_normal((1, ntok, cdim)).astype(np.float32)
y = rng.standard_normal((1, ydim)).astype(np.float32)
kw = dict(width=lat * 8, height=lat * 8, cfg=cfg, seed=42, batch=B, device_batch=B, method=sd.EULER if dit else sd.EULER_A, cond_y=y, uncond_y=y, fuse_cfg=True, device_sampler=True)
unc = None if nfwd == 1 else uncond
res = {v: [] for v in vals}
outs = {}
for r in range(rounds):
    for v in vals:
        sd.backend_set_option(key, v)
        outs[v] = eng.sample_latents(cond, unc, steps=2, **kw)  # plan + capture outside the timed region
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sample_latents(cond, unc, steps=steps, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        res[v].append(dt)
        print(f"round {r} {key}={v}: {dt:8.3f} ms/step", flush=True)
for v in vals:
    sd.backend_set_option(key, v)
    eng.sample_latents(cond, unc, steps=1, **kw)
    sd.kernel_timing_enable(sd.KF_ALL)
    eng.sample_latents(cond, unc, steps=1, **kw)
    fams = sorted(sd.kernel_timings(), key=lambda f: -f["total_ms"])
    sd.kernel_timing_enable(0)
    tot = sum(f["total_ms"] for f in fams)
    print(f"{key}={v}: median {np.median(res[v]):.3f} ms/step (min {min(res[v]):.3f}); kernels {tot:.2f} ms/step:")
    for f in fams[:8]:
        print(f"      {f['kernel'][:70]:70s} {f['launches']:5d} launches {f['total_ms']:8.3f} ms")
base = outs[vals[0]]
for v in vals[1:]:
    d = float(np.abs(outs[v] - base).max())
    print(f"latents after 2 steps, {key}={v} vs {key}={vals[0]}: max |diff| {d:.3e} (rel-L2 {float(np.linalg.norm(outs[v] - base) / np.linalg.norm(base)):.2e})")
