"""Probe: the reference runner's FLUX graph vs the engine's on the GPU (same topology): determinism, distance from the oracle, fusion counters."""
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
sd.load_backend(ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so")
name = sys.argv[1] if len(sys.argv) > 1 else "FLUX_TINY"
c = inputs_for(sd, name, np.random.default_rng(11))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64).ravel() - b.astype(np.float64).ravel()) / np.linalg.norm(b.astype(np.float64).ravel()))
e = sd.Engine(model=c["model"], backend="MI355X0", flash_attn=True)
eo = sd.Engine(model=c["model"], backend="CPU-oracle", flash_attn=True)
r = rg.RefRunner(e, c["family"], c["version"], "MI355X0", flash_attn=True, overrides=c["overrides"])
ora = c["eng"](eo)
keys = None
def delta(fn):
    s0 = sd.backend_stats(); o = fn(); s1 = sd.backend_stats()
    return o, {k: s1[k] - s0[k] for k in s1 if s1[k] != s0[k]}
for opt in ([], [("fusion", 0)], [("fuse_joint_qkv", 0)], [("fuse_rope", 0)], [("fuse_gate", 0)], [("fuse_cat_rows16", 0)], [("fuse_modulate", 0)], [("fuse_concat_heads", 0)], [("hip_graph", 0)]):
    for k, v in opt:
        sd.backend_set_option(k, v)
    a1, da = delta(lambda: r.compute(c["out"], **c["ref"]))
    a2, _ = delta(lambda: r.compute(c["out"], **c["ref"]))
    b1, db = delta(lambda: c["eng"](e))
    b2, _ = delta(lambda: c["eng"](e))
    print(opt, "ref det", np.array_equal(a1, a2), "eng det", np.array_equal(b1, b2), "ref==eng", np.array_equal(a1, b1.reshape(a1.shape)),
          "ref vs oracle %.2e eng vs oracle %.2e ref vs eng %.2e" % (rel(a1, ora.reshape(a1.shape)), rel(b1, ora), rel(a1, b1.reshape(a1.shape))))
    diff = {k: (da.get(k, 0), db.get(k, 0)) for k in set(da) | set(db) if da.get(k, 0) != db.get(k, 0)}
    print("   counters that differ (ref, eng):", diff)
    for k, v in opt:
        sd.backend_set_option(k, 1)
