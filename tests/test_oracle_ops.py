"""CPU suite (no GPU): pins the oracle (oracle/ggml_cpu_ref.cpp) against the committed PyTorch-fp32 golden vectors
(tests/golden/ops_torch_fp32.npz, produced by tests/golden/make_golden.py) for every op on the hot path, and checks the
ggml-cpu rounding points it is supposed to reproduce.  Tolerances: f32 ops 1e-5 abs; f16-weight contractions carry the
activation->f16 rounding (rel-L2 2e-3 vs exact-activation torch); GELU goes through the f16 table (2e-3)."""
from pathlib import Path

import numpy as np
import pytest

from ggml_graph import F16, F32, Q4_0, Q8_0, Graph, dequant

G = np.load(Path(__file__).resolve().parent / "golden" / "ops_torch_fp32.npz")


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def run(sd, dev, build):
    with Graph(dev) as g:
        return g.run(build(g, sd.lib()))


def test_linear(sd, oracle):
    def build(g, L):
        y = L.ggml_mul_mat(g.ctx, g.weight(G["lin_w"], F16), g.input(G["lin_x"]))
        return L.ggml_add_inplace(g.ctx, y, g.weight(G["lin_b"], F32))

    out = run(sd, oracle, build).reshape(G["lin_y"].shape)
    assert rel_l2(out, G["lin_y"]) < 2e-3
    # exact statement of the rounding point: activations rounded to f16, f32 accumulate
    xr = G["lin_x"].astype(np.float16).astype(np.float64)
    wr = G["lin_w"].astype(np.float16).astype(np.float64)
    assert np.abs(out - (xr @ wr.T + G["lin_b"])).max() < 2e-5


@pytest.mark.parametrize("key,stride", [("conv_y_s1", 1), ("conv_y_s2", 2)])
def test_conv3x3(sd, oracle, key, stride):
    def build(g, L):
        y = L.ggml_conv_2d(g.ctx, g.weight(G["conv_w"], F16), g.input(G["conv_x"]), stride, stride, 1, 1, 1, 1)
        return L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(G["conv_b"], F32), 1, 1, 20, 1))

    out = run(sd, oracle, build)
    assert out.shape == G[key].shape
    assert rel_l2(out, G[key]) < 2e-3

    def direct(g, L):
        y = L.ggml_conv_2d_direct(g.ctx, g.weight(G["conv_w"], F16), g.input(G["conv_x"]), stride, stride, 1, 1, 1, 1)
        return L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(G["conv_b"], F32), 1, 1, 20, 1))

    out2 = run(sd, oracle, direct)
    assert rel_l2(out2, out) < 1e-6


def test_conv1x1(sd, oracle):
    out = run(sd, oracle, lambda g, L: L.ggml_conv_2d(g.ctx, g.weight(G["conv1_w"], F16), g.input(G["conv_x"]), 1, 1, 0, 0, 1, 1))
    assert rel_l2(out, G["conv1_y"]) < 2e-3


def test_group_norm_affine_silu(sd, oracle):
    def build(g, L):
        t = L.ggml_group_norm(g.ctx, g.input(G["gn_x"]), 32, 1e-6)
        t = L.ggml_mul_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(G["gn_w"], F32), 1, 1, 64, 1))
        t = L.ggml_add_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(G["gn_b"], F32), 1, 1, 64, 1))
        return L.ggml_silu_inplace(g.ctx, t)

    assert np.abs(run(sd, oracle, build) - G["gn_y"]).max() < 2e-5


def test_layer_and_rms_norm(sd, oracle):
    def build(g, L):
        t = L.ggml_norm(g.ctx, g.input(G["ln_x"]), 1e-5)
        t = L.ggml_mul_inplace(g.ctx, t, g.weight(G["ln_w"], F32))
        return L.ggml_add_inplace(g.ctx, t, g.weight(G["ln_b"], F32))

    assert np.abs(run(sd, oracle, build).reshape(G["ln_y"].shape) - G["ln_y"]).max() < 2e-5
    out = run(sd, oracle, lambda g, L: L.ggml_rms_norm(g.ctx, g.input(G["ln_x"]), 1e-6))
    assert np.abs(out.reshape(G["rms_y"].shape) - G["rms_y"]).max() < 2e-5


@pytest.mark.parametrize("fn,key,tol", [("ggml_silu", "silu_y", 2e-6), ("ggml_gelu", "gelu_y", 2e-3), ("ggml_gelu_quick", "gelu_quick_y", 2e-3),
                                         ("ggml_sigmoid", "sigmoid_y", 2e-6)])
def test_activations(sd, oracle, fn, key, tol):
    out = run(sd, oracle, lambda g, L: getattr(L, fn)(g.ctx, g.input(G["act_x"])))
    assert np.abs(out.reshape(G[key].shape) - G[key]).max() <= tol * max(1.0, np.abs(G[key]).max())


def test_softmax(sd, oracle):
    out = run(sd, oracle, lambda g, L: L.ggml_soft_max(g.ctx, g.input(G["sm_x"])))
    assert np.abs(out.reshape(G["sm_y"].shape) - G["sm_y"]).max() < 1e-6


def test_attention_both_encodings(sd, oracle):
    q, k, v = G["att_q"], G["att_k"], G["att_v"]
    scale = 1.0 / np.sqrt(q.shape[-1])

    def manual(g, L):  # ggml_extend.hpp:1460-1479
        kq = L.ggml_mul_mat(g.ctx, g.input(k), g.input(q))
        kq = L.ggml_soft_max_inplace(g.ctx, L.ggml_scale_inplace(g.ctx, kq, scale))
        vt = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, g.input(v), 1, 0, 2, 3))
        return L.ggml_mul_mat(g.ctx, vt, kq)

    out = run(sd, oracle, manual)
    assert rel_l2(out.reshape(G["att_y"].shape), G["att_y"]) < 1e-5   # exact f32 path

    def flash(g, L):  # ggml_extend.hpp:1396-1431
        return L.ggml_flash_attn_ext(g.ctx, g.input(q), g.input(k, F16), g.input(v, F16), None, scale, 0.0, 0.0)

    out = run(sd, oracle, flash)  # [1, Lq, HN, d]
    # K/V/Q rounded to f16 and V accumulated in f16 (Appendix E.3): ~1e-3 class
    assert rel_l2(out[0].transpose(1, 0, 2), G["att_y"]) < 5e-3


def test_upscale_concat_timestep(sd, oracle):
    out = run(sd, oracle, lambda g, L: L.ggml_upscale(g.ctx, g.input(G["up_x"]), 2, 0))
    np.testing.assert_array_equal(out, G["up_y"])
    a, b = G["conv_x"], G["conv_x"][:, :5]
    out = run(sd, oracle, lambda g, L: L.ggml_concat(g.ctx, g.input(a), g.input(b), 2))
    np.testing.assert_array_equal(out, np.concatenate([a, b], 1))
    out = run(sd, oracle, lambda g, L: L.ggml_timestep_embedding(g.ctx, g.input(G["te_t"]), 320, 10000))
    assert np.abs(out.reshape(G["te_y"].shape) - G["te_y"]).max() < 2e-4   # f32 freq * t vs float64 reference at t=999


@pytest.mark.parametrize("wtype,tol", [(Q8_0, 1.5e-2), (Q4_0, 1.2e-1)])
def test_quantized_weights_rounding_points(sd, oracle, wtype, tol):
    """Q8_0/Q4_0 weights: ggml-cpu quantises the ACTIVATIONS to Q8_0 too (Appendix E.1).  Check the oracle against an
    independent numpy statement of exactly that arithmetic, and that the result is within quantisation noise of torch."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((9, 64)).astype(np.float32)
    w = (rng.standard_normal((12, 64)) / 8).astype(np.float32)
    out = run(sd, oracle, lambda g, L: L.ggml_mul_mat(g.ctx, g.weight(w, wtype), g.input(x))).reshape(9, 12)
    wq = dequant(w, wtype).astype(np.float64)
    # activation Q8_0 blocks: d = amax/127 (stored f16), q = round(x/d)
    xb = x.reshape(9, 2, 32)
    d = np.abs(xb).max(-1, keepdims=True) / 127.0
    qx = np.where(d > 0, np.round(xb / np.where(d > 0, d, 1)), 0)
    xq = (qx * d.astype(np.float16).astype(np.float32)).reshape(9, 64).astype(np.float64)
    assert np.abs(out - xq @ wq.T).max() < 1e-4
    assert rel_l2(out, x.astype(np.float64) @ w.astype(np.float64).T) < tol
