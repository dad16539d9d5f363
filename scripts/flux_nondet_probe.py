"""Probe: which node of the FLUX_TINY forward differs between PROCESSES on the GPU?  Prints a checksum per callback tensor (cut behind every node of a given op set)."""
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
name = sys.argv[1] if len(sys.argv) > 1 else "FLUX_TINY"
every = len(sys.argv) > 2 and sys.argv[2] == "all"
c = inputs_for(sd, name, np.random.default_rng(11))
e = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=True)
if len(sys.argv) > 3:  # a second backend instance on the same device, used first
    e2 = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=True)
    print("E2", hashlib.sha1(c["eng"](e2).tobytes()).hexdigest()[:12])
whole = c["eng"](e)
print("WHOLE", hashlib.sha1(whole.tobytes()).hexdigest()[:12])
mm = sd.op_number("MUL_MAT")
noop = {sd.op_number(n) for n in ("RESHAPE", "VIEW", "PERMUTE", "TRANSPOSE", "NONE")}
with sd.EvalTrace((lambda i, ts: ts.op not in noop) if every else (lambda i, ts: ts.op == mm)) as tr:
    out = c["eng"](e)
print("SLICED", hashlib.sha1(out.tobytes()).hexdigest()[:12])
for r in tr.records:
    h = hashlib.sha1(r[3].tobytes()).hexdigest()[:10] if r[3] is not None else "-"
    h1 = hashlib.sha1(r[4].tobytes()).hexdigest()[:10] if r[4] is not None else "-"
    print("REC", r[0], r[1], r[2][:30], h, h1)
