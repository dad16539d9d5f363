"""GPU leg of the enum-drift defence (VERDICT r4 task 4; reference call sites src/core/ggml_extend_backend.cpp:302-320, 466-509).

The same graphs run through two HOSTS and one UNCHANGED plug-in:
  * libsdcpp-host.so          — `enum ggml_op` numbered like include/ggml-abi.h,
  * libsdcpp-host-opshift.so  — a "fork" with two ops and one unary op inserted mid-enum (-DGGML_ABI_TEST_SHIFTED_ENUMS).
ggml_backend_init() has to find the fork's numbering through the host's ggml_op_name() / ggml_unary_op_name(); every kernel choice is then
the same, so the outputs must be bit-identical.  (The CPU oracle plug-in is compiled against the unshifted header and cannot serve the
shifted host: the shifted leg is compared with the unshifted GPU leg, which the rest of the suite compares with the oracle.)"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

_WORKER = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import sdcpp_amd as sd
sd.load_mi355x_backend()
be = sd._backend()
import ctypes as C
be.ggml_backend_mi355x_enum_status.restype = C.c_char_p
ops = (C.c_uint8 * 256)(); un = (C.c_uint8 * 256)()
be.ggml_backend_mi355x_get_enum_maps(ops, un)
rng = np.random.default_rng(5)
out = {}
x = rng.standard_normal((2, 4, 32, 32)).astype(np.float32)
t = np.array([731.0, 100.0], dtype=np.float32)
ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
out["unet"] = sd.Engine(model=sd.SD15_TINY, backend="MI355X0", flash_attn=True).unet_forward(x, t, ctx)
out["vae"] = sd.Engine(model=sd.SD15_TINY, backend="MI355X0").vae_decode(x[:1] * 0.5)
xf = rng.standard_normal((2, 16, 18, 15)).astype(np.float32)
tf = np.array([0.81, 0.27], dtype=np.float32)
cf = rng.standard_normal((1, 40, 96)).astype(np.float32)
yf = rng.standard_normal((1, 64)).astype(np.float32)
out["flux"] = sd.Engine(model=sd.FLUX_TINY, backend="MI355X0", flash_attn=True, wtype=sd.Q4_0).unet_forward(xf, tf, cf, yf)
out["mmdit"] = sd.Engine(model=sd.SD35_TINY, backend="MI355X0", flash_attn=False).unet_forward(xf, np.array([731.0, 210.0], np.float32), np.concatenate([cf] * 2, 1), yf)
np.savez(sys.argv[2], **out)
st = sd.backend_stats()
print(json.dumps({"status": be.ggml_backend_mi355x_enum_status().decode(), "identity": all(ops[i] in (i, ops[255]) for i in range(100)) and sum(ops[i] == i for i in range(100)) >= 25,   # ops[255] = "unknown to this backend"
                  "fused_conv": st["fused_conv"], "fused_rope": st["fused_rope"], "fused_attention": st["fused_attention"], "generic_matmul": st["generic_matmul"]}))
"""


def _run(host_so, out_npz):
    import json

    env = dict(os.environ, SDCPP_HOST_LIB=str(host_so))
    r = subprocess.run([sys.executable, "-c", _WORKER, str(ROOT), str(out_npz)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1]), np.load(out_npz)


def test_same_graphs_through_a_host_with_shifted_op_numbers(sd, gpu, tmp_path):
    if gpu != "MI355X0":
        pytest.skip("needs the HIP backend")
    base_info, base = _run(sd.LIB_DIR / "libsdcpp-host.so", tmp_path / "base.npz")
    fork_info, fork = _run(sd.LIB_DIR / "libsdcpp-host-opshift.so", tmp_path / "fork.npz")
    print(base_info, fork_info)
    assert "translated by name" in base_info["status"] and base_info["identity"]
    assert "translated by name" in fork_info["status"] and not fork_info["identity"]
    for k in ("fused_conv", "fused_rope", "fused_attention", "generic_matmul"):  # the same patterns were recognised
        assert base_info[k] == fork_info[k], k
    assert base_info["fused_conv"] > 0 and base_info["fused_rope"] > 0
    for k in base.files:
        assert np.isfinite(base[k]).all()
        np.testing.assert_array_equal(base[k], fork[k], err_msg=k)
