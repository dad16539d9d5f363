"""Probe: with two backend instances on one device, which planner option removes the run-to-run differences of the second instance's first FLUX_TINY forward?"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sdcpp_amd as sd
from test_ref_graphs import inputs_for

sd.load_mi355x_backend()
model = sys.argv[1]
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    sd.backend_set_option(k, int(v))
c = inputs_for(sd, model, np.random.default_rng(11))
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
e = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=True)
e2 = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=True)
a = h(c["eng"](e2))
b = h(c["eng"](e))
print(model, sys.argv[2:], "SAME" if a == b else "DIFF", a, b)
