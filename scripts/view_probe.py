"""Probe: how many nodes of a sub-graph view does the planner treat as externally visible, and how far is a sliced run from the whole-graph run?"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sdcpp_amd as sd

sd.load_mi355x_backend()
rng = np.random.default_rng(1)
x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
t = np.array([731.0, 210.0], dtype=np.float32)
ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
e = sd.Engine(model=sd.SD15_TINY, backend="MI355X0", flash_attn=True)
whole = e.unet_forward(x, t, ctx)
whole2 = e.unet_forward(x, t, ctx)
print("whole vs whole again identical:", np.array_equal(whole, whole2))


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))


for label, want in (("one slice = whole graph as a view", lambda i, ts: False), ("cut at node 1000 only", lambda i, ts: i == 1000),
                    ("cut at 20 nodes", lambda i, ts: i % 128 == 127)):
    s0 = sd.backend_stats()
    with sd.EvalTrace(want) as tr:
        out = e.unet_forward(x, t, ctx)
    s1 = sd.backend_stats()
    d = {k: s1[k] - s0[k] for k in ("view_graphs", "view_external_nodes", "nodes_seen", "kernels_planned", "fused_conv", "fused_linear", "fused_attention", "fused_norm")}
    print(f"{label}: vs whole {rel(out, whole):.2e} identical {np.array_equal(out, whole)} {d}")
for opt in ("hoist_kv", "hoist_emb", "fuse_siblings"):
    sd.backend_set_option(opt, 0)
    o2 = e.unet_forward(x, t, ctx)
    print(f"whole graph with {opt}=0 vs default: {rel(o2, whole):.2e}")
    sd.backend_set_option(opt, 1)
