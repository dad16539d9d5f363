"""Latency of the text-encoder step at real width on the GPU: python scripts/te_bench.py [sd15|sdxl ...]
Prints one JSON line per model: ms per get_learned_condition call (77 tokens, synthetic weights and ids), after one warm-up call
(the warm-up builds the weight images and the plan cache)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np
import torch  # noqa: F401  (initialise HIP before the backend does)

import sdcpp_amd as sd

L = 77
for name in sys.argv[1:] or ["sd15"]:
    model = {"sd15": sd.SD15, "sdxl": sd.SDXL, "sd35": sd.SD35_LARGE, "flux": sd.FLUX_DEV}[name]
    t0 = time.time()
    e = sd.Engine(model=model, wtype=sd.F16)
    e.text_encoders_init()
    t_init = time.time() - t0
    rng = np.random.default_rng(0)
    ids = np.full(L, 49407, np.int32)
    ids[0] = 49406
    ids[1:12] = rng.integers(1000, 40000, 11)
    kw = dict(width=1024, height=1024)
    args = (ids,) if name in ("sd15", "sdxl") else (ids, ids, rng.integers(0, 32000, L if name == "sd35" else 256).astype(np.int32))
    e.get_learned_condition(*args, **kw)
    t0 = time.time()
    n = 5
    for _ in range(n):
        c, y = e.get_learned_condition(*args, **kw)
    ms = (time.time() - t0) / n * 1e3
    print(json.dumps({"model": name, "ms_per_condition": round(ms, 2), "ctx_shape": list(c.shape), "y_dim": None if y is None else int(y.shape[1]),
                      "init_s": round(t_init, 1), "finite": bool(np.isfinite(c).all())}), flush=True)
    e.close()
