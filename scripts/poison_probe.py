"""Probe (GGML_MI355X_POISON=1): every backend buffer and the operand arena start as NaN patterns; a kernel reading memory nothing wrote shows up as NaN.
Runs every tiny model whole and sliced behind every MUL_MAT, reports the first callback tensor that is not finite."""
import os
import sys
from pathlib import Path

import numpy as np

os.environ["GGML_MI355X_POISON"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from test_ref_graphs import inputs_for

sd.load_mi355x_backend()
mm = sd.op_number("MUL_MAT")
for name in ["SD15_TINY", "SDXL_TINY", "VAE", "VAE_SDXL", "VAE16", "SD35_TINY", "SD3M_TINY", "FLUX_TINY"]:
    for flash in (True, False):
        c = inputs_for(sd, name, np.random.default_rng(11))
        e = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=flash)
        whole = c["eng"](e)
        again = c["eng"](e)
        with sd.EvalTrace(lambda i, ts: ts.op == mm) as tr:
            sliced = c["eng"](e)
        bad = [(r[0], r[2], "node" if not np.isfinite(r[3]).all() else "src1") for r in tr.records
               if (r[3] is not None and not np.isfinite(r[3]).all()) or (r[4] is not None and not np.isfinite(r[4]).all())]
        print(f"{name} flash={flash}: whole finite {np.isfinite(whole).all()} (nan count {int(np.isnan(whole).sum())}), replay identical {np.array_equal(whole, again, equal_nan=True)}, "
              f"sliced finite {np.isfinite(sliced).all()}, first bad callback tensors {bad[:3]}", flush=True)
