"""Probe: does a second backend instance on the same device change the engine's FLUX_TINY result?"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
import ref_graphs as rg
from test_ref_graphs import inputs_for

sd.load_mi355x_backend()
mode = sys.argv[1]
c = inputs_for(sd, "FLUX_TINY", np.random.default_rng(11))
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
e = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=True)
out = []
if mode == "engine_only":
    out.append(("eng", h(c["eng"](e))))
elif mode == "second_engine":
    e2 = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=True)
    out.append(("eng2", h(c["eng"](e2))))
    out.append(("eng", h(c["eng"](e))))
elif mode == "ref_created":
    r = rg.RefRunner(e, c["family"], c["version"], "MI355X0", flash_attn=True, overrides=c["overrides"])
    out.append(("eng", h(c["eng"](e))))
elif mode == "ref_first":
    r = rg.RefRunner(e, c["family"], c["version"], "MI355X0", flash_attn=True, overrides=c["overrides"])
    out.append(("ref", h(r.compute(c["out"], **c["ref"]))))
    out.append(("eng", h(c["eng"](e))))
    out.append(("ref", h(r.compute(c["out"], **c["ref"]))))
    out.append(("eng", h(c["eng"](e))))
elif mode == "eng_first":
    r = rg.RefRunner(e, c["family"], c["version"], "MI355X0", flash_attn=True, overrides=c["overrides"])
    out.append(("eng", h(c["eng"](e))))
    out.append(("ref", h(r.compute(c["out"], **c["ref"]))))
print(mode, out)
