#!/usr/bin/env python
"""Within-process interleaved A/B of a backend option on the bench workload (SD1.5 512x512, fused cfg pair, batch 8).
usage: ab_bench.py <option> <v0,v1,...> [rounds] [steps]   e.g.  ab_bench.py conv_tap_major 0,1 3 4"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import sdcpp_amd as sd

key = sys.argv[1]
vals = [int(v) for v in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
B = 8
sd.load_mi355x_backend()
eng = sd.Engine(model=sd.SD15, backend="MI355X0", wtype=sd.F16, flash_attn=True)
rng = np.random.default_rng(0)
x2 = rng.standard_normal((2 * B, 4, 64, 64)).astype(np.float32)
t2 = np.full((2 * B,), 500.0, dtype=np.float32)
c2 = rng.standard_normal((2, 77, 768)).astype(np.float32)
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        sd.backend_set_option(key, v)
        eng.unet_forward(x2, t2, c2, None)  # rebuild plan (+ weight images) outside the timed region
        eng.unet_forward(x2, t2, c2, None)
        sd.kernel_timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.unet_forward(x2, t2, c2, None)
        kt = sd.kernel_timing()
        dt = (time.perf_counter() - t0) / steps * 1e3
        sd.kernel_timing_enable(False)
        res[v].append((dt, kt["total_ms"] * 1e3 / max(kt["launches"], 1), kt["total_flops"] / max(kt["total_ms"], 1e-9) / 1e9))
        print(f"round {r} {key}={v}: step {dt:7.2f} ms   dominant conv avg {res[v][-1][1]:7.1f} us  {res[v][-1][2]:7.1f} TF/s", flush=True)
for v in vals:
    a = np.array(res[v])
    print(f"{key}={v}: median step {np.median(a[:,0]):.2f} ms (min {a[:,0].min():.2f});  conv median {np.median(a[:,1]):.1f} us (min {a[:,1].min():.1f})")
