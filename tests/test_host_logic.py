"""CPU suite: host-side pieces of the hot path (no arithmetic backend involved) — sampler known-answer tests, block
encoders, graph front-end + allocator invariants."""
import ctypes as C

import numpy as np
import pytest

from ggml_graph import F16, F32, Q4_0, Q8_0, Graph, dequant, encode, tensor_struct


# ---- independent numpy Philox4x32-10 (written from the Random123 / cuRAND definition, not from the engine) ----
def philox_randn_np(seed: int, offset: int, n: int) -> np.ndarray:
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = np.zeros((4, n), dtype=np.uint64)
    c[0] = offset
    c[2] = np.arange(n, dtype=np.uint64)
    k0 = np.full(n, seed & 0xFFFFFFFF, dtype=np.uint64)
    k1 = np.full(n, (seed >> 32) & 0xFFFFFFFF, dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for r in range(10):
        p0 = c[0] * np.uint64(M0)
        p1 = c[2] * np.uint64(M1)
        c = np.stack([(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & mask])
        if r < 9:
            k0 = (k0 + np.uint64(W0)) & mask
            k1 = (k1 + np.uint64(W1)) & mask
    f32 = np.float32
    u = c[0].astype(np.float32) * f32(2.3283064e-10) + f32(2.3283064e-10) / f32(2)
    v = c[1].astype(np.float32) * (f32(2.3283064e-10) * f32(6.2831855)) + (f32(2.3283064e-10) * f32(6.2831855)) / f32(2)
    return (np.sqrt(f32(-2.0) * np.log(u)) * np.sin(v)).astype(np.float32)


def test_philox_known_answers(sd):
    # rng_philox.hpp:101-122: counter=(offset,0,i,0), key=seed, Box-Muller on the first two words
    for seed, offset in [(42, 0), (42, 1), (2**33 + 7, 5), (0, 0)]:
        a = sd.philox_randn(seed, offset, 4096)
        b = philox_randn_np(seed, offset, 4096)
        assert np.abs(a - b).max() < 5e-6
    a = sd.philox_randn(42, 0, 200000)
    assert abs(a.mean()) < 0.01 and abs(a.std() - 1.0) < 0.01   # it is a standard normal


def test_sigma_ladder_known_answers(sd):
    # k-diffusion's published SD1.x constants: sigma_max 14.6146, sigma_min 0.0292 (scaled-linear betas 0.00085..0.012)
    s = sd.get_sigmas(20)
    assert len(s) == 21 and s[-1] == 0.0
    assert abs(s[0] - 14.6146) < 2e-3
    assert np.all(np.diff(s) < 0)
    s1000 = sd.get_sigmas(1000)
    assert abs(s1000[999] - 0.0292) < 2e-4
    # sigma_to_t inverts t_to_sigma on the ladder (denoiser.hpp:1140-1172)
    L = sd.lib()
    ts = [L.sd_sigma_to_t(float(v)) for v in s[:-1]]
    np.testing.assert_allclose(ts, np.linspace(999, 0, 20), atol=2e-2)


def test_quant_block_encoders(sd):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((4, 64)).astype(np.float32)
    # Q8_0: |x - dq| <= d/2 with d = amax/127 per 32-block (Appendix D)
    dq = dequant(x, Q8_0)
    d = np.abs(x.reshape(4, 2, 32)).max(-1, keepdims=True) / 127
    assert np.all(np.abs(x.reshape(4, 2, 32) - dq.reshape(4, 2, 32)) <= d * 0.57 + 1e-6)  # 0.5 d rounding + f16 rounding of d (127 * 2^-11)
    raw = encode(x, Q8_0)
    assert len(raw) == 4 * 2 * 34
    # Q4_0: 18-byte blocks, values on the 16-level grid (q-8)*d
    raw4 = encode(x, Q4_0)
    assert len(raw4) == 4 * 2 * 18
    dq4 = dequant(x, Q4_0).reshape(8, 32)
    for blk, xb in zip(dq4, x.reshape(8, 32)):
        m = xb[np.argmax(np.abs(xb))]
        dd = np.float32(np.float16(m / -8.0))
        q = blk / dd + 8
        assert np.allclose(q, np.round(q), atol=1e-3) and q.min() >= -1e-3 and q.max() <= 15 + 1e-3


def test_graph_builder_conv_chain_shapes(sd):
    """ggml_conv_2d expands to IM2COL(F16) -> MUL_MAT -> RESHAPE -> PERMUTE -> CONT."""
    L = sd.lib()
    ctx = L.ggml_init(sd.GgmlInitParams(0, None, True))
    x = L.ggml_new_tensor_4d(ctx, F32, 16, 12, 8, 2)
    w = L.ggml_new_tensor_4d(ctx, F16, 3, 3, 8, 20)
    y = L.ggml_conv_2d(ctx, w, x, 2, 2, 1, 1, 1, 1)
    ys = tensor_struct(y)
    assert [ys.ne[i] for i in range(4)] == [8, 6, 20, 2]
    gf = L.ggml_new_graph(ctx)
    L.ggml_build_forward_expand(gf, y)
    ops = [tensor_struct(L.ggml_graph_node(gf, i)).op for i in range(L.ggml_graph_n_nodes(gf))]
    names = {52: "IM2COL", 36: "RESHAPE", 29: "MUL_MAT", 38: "PERMUTE", 35: "CONT"}
    assert [names[o] for o in ops] == ["IM2COL", "RESHAPE", "RESHAPE", "MUL_MAT", "RESHAPE", "PERMUTE", "CONT"]
    im = tensor_struct(L.ggml_graph_node(gf, 0))
    assert im.type == F16 and [im.ne[i] for i in range(4)] == [72, 8, 6, 2]
    L.ggml_free(ctx)


def test_gallocr_is_deterministic_and_recycles(sd, oracle):
    """Same topology -> same addresses (what the backend's plan cache relies on), and dead tensors are reused."""
    L = sd.lib()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 64, 16, 16)).astype(np.float32)
    offs = []
    sizes = []
    for _ in range(2):
        with Graph(oracle) as g:
            t = g.input(x)
            nodes = []
            for _ in range(6):
                t = L.ggml_silu(g.ctx, L.ggml_scale(g.ctx, t, 1.01))
                nodes.append(t)
            out = g.run(t)
            base = min(tensor_struct(n).data for n in nodes)
            offs.append([tensor_struct(n).data - base for n in nodes])
            sizes.append(L.ggml_gallocr_get_buffer_size(g._galloc, 0))
    assert offs[0] == offs[1] and sizes[0] == sizes[1]
    # 12 intermediate tensors of 131072 B each would need 1.5 MiB without reuse
    assert sizes[0] <= 4 * x.nbytes
    np.testing.assert_allclose(out, _ref_chain(x), rtol=1e-5)


def _ref_chain(x):
    x = x.astype(np.float64)
    for _ in range(6):
        x = x * 1.01
        x = x / (1 + np.exp(-x))
    return x


def test_inplace_ops_alias_their_source(sd):
    L = sd.lib()
    ctx = L.ggml_init(sd.GgmlInitParams(0, None, True))
    a = L.ggml_new_tensor_2d(ctx, F32, 8, 4)
    b = L.ggml_new_tensor_1d(ctx, F32, 8)
    c = L.ggml_add_inplace(ctx, a, b)
    assert tensor_struct(c).view_src == a
    v = L.ggml_permute(ctx, a, 1, 0, 2, 3)
    vs = tensor_struct(v)
    assert (vs.ne[0], vs.ne[1], vs.nb[0], vs.nb[1]) == (4, 8, 32, 4)
    L.ggml_free(ctx)
