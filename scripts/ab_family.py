#!/usr/bin/env python
"""Within-process interleaved A/B of launch-time backend options on one model forward, per kernel family (HIP events around every dispatch).
usage: ab_family.py <sd15|sdxl|sd35|flux> [reps] -- key=int[,key=int...] -- key=int[,...]   (option sets A and B; run A B A B ...)
Only options read at LAUNCH time can be compared this way (tile policy: tail_split, t256p_pad, gemm16_tile, streamk ...): plans are cached per graph."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import sdcpp_amd as sd

sd.load_mi355x_backend()
rng = np.random.default_rng(0)
argv = sys.argv[1:]
what = argv[0]
reps = int(argv[1]) if len(argv) > 1 and argv[1].isdigit() else 3
sets = [[kv.split("=") for kv in part.split(",") if kv] for part in " ".join(argv).split("--")[1:]]
sets = [[(k.strip(), int(v)) for k, v in s] for s in sets]
if what == "sd15":
    e = sd.Engine(model=sd.SD15, flash_attn=True)
    args = (rng.standard_normal((16, 4, 64, 64)).astype(np.float32), np.full(16, 500.0, np.float32), rng.standard_normal((16, 77, 768)).astype(np.float32), None)
elif what == "sdxl":
    e = sd.Engine(model=sd.SDXL, wtype=sd.Q8_0, flash_attn=True)
    args = (rng.standard_normal((2, 4, 128, 128)).astype(np.float32), np.full(2, 500.0, np.float32), rng.standard_normal((2, 77, 2048)).astype(np.float32),
            rng.standard_normal((2, 2816)).astype(np.float32))
else:
    flux = what == "flux"
    e = sd.Engine(model=sd.FLUX_DEV if flux else sd.SD35_LARGE, wtype=sd.Q4_0 if flux else sd.BF16, flash_attn=True)
    n = 1 if flux else 2
    args = (rng.standard_normal((n, 16, 128, 128)).astype(np.float32), np.full(n, 0.5 if flux else 500.0, np.float32),
            rng.standard_normal((1 if flux else n, 256 if flux else 154, 4096)).astype(np.float32), rng.standard_normal((n, 768 if flux else 2048)).astype(np.float32))
e.unet_forward(*args)
res = [[] for _ in sets]
for r in range(reps):
    for i, s in enumerate(sets):
        for k, v in s:
            sd.backend_set_option(k, v)
        e.unet_forward(*args)
        sd.kernel_timing_enable(sd.KF_ALL)
        e.unet_forward(*args)
        fams = sd.kernel_timings()
        sd.kernel_timing_enable(0)
        tot = sum(f["total_ms"] for f in fams)
        lin = [f for f in fams if f["kernel"].startswith("Linear")]
        res[i].append((tot, lin[0]["total_ms"] if lin else 0.0, (lin[0]["total_flops"] / (lin[0]["total_ms"] * 1e-3) / 1e12) if lin else 0.0, lin[0]["launches"] if lin else 0))
for i, s in enumerate(sets):
    t = sorted(x[0] for x in res[i])
    l = sorted(x[1] for x in res[i])
    tf = sorted(x[2] for x in res[i])
    print(f"{what} {dict(s)}: kernels per forward median {t[len(t) // 2]:.3f} ms (min {t[0]:.3f}); Linear family median {l[len(l) // 2]:.3f} ms = {tf[len(tf) // 2]:.1f} TFLOP/s, {res[i][0][3]} launches", flush=True)
