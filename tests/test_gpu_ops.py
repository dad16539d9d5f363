"""Op-level parity: the SAME ggml graph is run through the CPU oracle plug-in and through libggml-mi355x.so
(the C-ABI backend vtables) on seeded inputs.  Tolerances are stated per test:
  * memory-bound f32 ops: max-abs <= 2e-5 relative to the output scale (f32 rounding / fast-exp class);
  * f16-operand contractions (MUL_MAT with f16 weight, conv): the oracle rounds activations to f16 exactly like
    the MFMA path does, so only f32 summation order differs: rel-L2 <= 2e-4;
  * q8_0 / q4_0 weights: the oracle (ggml-cpu) ALSO quantises activations to q8_0, the GPU keeps them f16
    => GPU is closer to the exact product; rel-L2 vs oracle <= 1e-2 (q8_0) / 3e-2 (q4_0) as SURVEY.md suggests,
    and rel-L2 vs the exact dequantised-weight product <= 2e-3.
"""
import os

import numpy as np
import pytest

from ggml_graph import BF16, F16, F32, I32, Q4_0, Q8_0, Graph, dequant

pytestmark = pytest.mark.gpu


def _on_gpu():
    """False in the harness self-check mode (SDCPP_GPU_TESTS_ON_ORACLE=1: the 'gpu' fixture is the CPU oracle)"""
    import os
    return os.environ.get("SDCPP_GPU_TESTS_ON_ORACLE") != "1"


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def run_both(sd, oracle, gpu, build):
    outs = []
    for dev in (oracle, gpu):
        with Graph(dev) as g:
            node = build(g, sd.lib())
            outs.append(g.run(node))
    return outs


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(1234)


@pytest.mark.parametrize("shape_a,shape_b", [
    ((2, 8, 16, 16), (2, 8, 16, 16)),      # same shape
    ((2, 8, 16, 16), (1, 8, 1, 1)),        # [1,1,C,1] channel broadcast (conv bias / GN affine)
    ((2, 8, 16, 16), (2, 8, 1, 1)),        # [1,1,C,N] (time-embedding add)
    ((1, 3, 77, 64), (64,)),               # row vector (linear bias)
    ((3, 5, 7, 9), (1, 5, 1, 9)),          # generic broadcast, odd sizes
])
@pytest.mark.parametrize("op", ["ggml_add", "ggml_mul", "ggml_sub", "ggml_div"])
def test_binary(sd, oracle, gpu, rng, shape_a, shape_b, op):
    a = rng.standard_normal(shape_a).astype(np.float32)
    b = (rng.standard_normal(shape_b).astype(np.float32) + 3.0)

    def build(g, L):
        return getattr(L, op)(g.ctx, g.input(a), g.input(b))

    ref, out = run_both(sd, oracle, gpu, build)
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("fn,tol", [("ggml_silu", 2e-6), ("ggml_gelu", 2e-3), ("ggml_gelu_quick", 2e-3), ("ggml_sigmoid", 2e-6),
                                     ("ggml_tanh", 2e-6), ("ggml_relu", 0)])
def test_unary(sd, oracle, gpu, rng, fn, tol):
    # GELU: the oracle goes through ggml-cpu's f16 table (input and output rounded to f16) -> 2e-3 abs at |x|<=6
    x = (rng.standard_normal((2, 5, 33, 20)) * 2).astype(np.float32)
    ref, out = run_both(sd, oracle, gpu, lambda g, L: getattr(L, fn)(g.ctx, g.input(x)))
    assert np.abs(out - ref).max() <= tol * max(1.0, np.abs(ref).max()) + 1e-7


def test_scale_and_timestep_embedding(sd, oracle, gpu, rng):
    x = rng.standard_normal((3, 1000)).astype(np.float32)
    ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_scale(g.ctx, g.input(x), 0.125))
    np.testing.assert_allclose(out, ref, rtol=1e-6)
    t = np.array([999.0, 500.25, 0.0, 13.5], dtype=np.float32)
    for dim in (320, 256, 33):
        ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_timestep_embedding(g.ctx, g.input(t), dim, 10000))
        # cos/sin of arguments up to 999: f32 argument-reduction differences ~1e-4 abs
        assert np.abs(out - ref).max() < 5e-4


@pytest.mark.parametrize("shape", [(2, 320, 16, 16), (1, 64, 8, 8), (2, 96, 5, 7), (1, 128, 64, 64)])
def test_group_norm_chain(sd, oracle, gpu, rng, shape):
    x = (rng.standard_normal(shape) * 3 + 0.5).astype(np.float32)
    C = shape[1]
    w = rng.standard_normal(C).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)

    def build(g, L):
        t = L.ggml_group_norm(g.ctx, g.input(x), 32, 1e-6)
        t = L.ggml_mul_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(w, F32), 1, 1, C, 1))
        t = L.ggml_add_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, C, 1))
        return L.ggml_silu_inplace(g.ctx, t)

    ref, out = run_both(sd, oracle, gpu, build)
    assert np.abs(out - ref).max() < 3e-5 * max(1.0, np.abs(ref).max())
    ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_group_norm(g.ctx, g.input(x), 32, 1e-6))
    assert np.abs(out - ref).max() < 3e-5


@pytest.mark.parametrize("shape,mean", [((1, 64, 256, 256), 5.0), ((2, 64, 64, 64), -2.0), ((1, 32, 300, 260), 0.3)])
def test_group_norm_into_conv_large_groups(sd, oracle, gpu, rng, shape, mean):
    """GN -> affine -> SiLU feeding a conv: statistics kernel + apply-and-pack kernel (no f32 normalised tensor).  Groups of 131072 / 8192 /
    78000 values with a mean far from zero: the one-pass statistics (sums relative to the group's first element) must match the oracle's
    double-precision two-pass mean / variance."""
    N, C, H, W = shape
    x = (rng.standard_normal(shape) * 1.5 + mean).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    wc = (rng.standard_normal((32, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)

    def build(g, L):
        t = L.ggml_group_norm(g.ctx, g.input(x), 32, 1e-6)
        t = L.ggml_mul_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(w, F32), 1, 1, C, 1))
        t = L.ggml_add_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, C, 1))
        t = L.ggml_silu_inplace(g.ctx, t)
        return L.ggml_conv_2d(g.ctx, g.weight(wc, F16), t, 1, 1, 1, 1, 1, 1)

    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all() and rel_l2(out, ref) < 3e-4


@pytest.mark.parametrize("N,C,inner,H,W", [(2, 640, 640, 16, 16), (1, 64, 96, 5, 8), (2, 1280, 1280, 32, 32)])
def test_group_norm_into_token_linear(sd, oracle, gpu, rng, N, C, inner, H, W):
    """SpatialTransformer with Linear projections (SDXL, block.hpp:548-566): GroupNorm (+affine) -> PERMUTE(1,2,0,3) -> CONT -> RESHAPE [C, W*H, N] -> proj_in Linear.
    The GroupNorm apply pass writes the Linear's f16 operand image: no f32 norm kernel, no transposing copy, no pack pass."""
    x = (rng.standard_normal((N, C, H, W)) * 1.2 + 0.4).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    wl = (rng.standard_normal((inner, C)) / np.sqrt(C)).astype(np.float32)
    bl = rng.standard_normal(inner).astype(np.float32)

    def build(g, L):
        t = L.ggml_group_norm(g.ctx, g.input(x), 32, 1e-6)
        t = L.ggml_mul_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(w, F32), 1, 1, C, 1))
        t = L.ggml_add_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, C, 1))
        t = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, t, 1, 2, 0, 3))
        t = L.ggml_reshape_3d(g.ctx, t, C, W * H, N)
        y = L.ggml_mul_mat(g.ctx, g.weight(wl, F16), t)
        return L.ggml_add_inplace(g.ctx, y, g.weight(bl, F32))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all() and rel_l2(out, ref) < 3e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS") and H * W % 4 == 0:
        assert sd.backend_stats()["fused_proj_tokens"] - before["fused_proj_tokens"] == 1


@pytest.mark.parametrize("N,C,inner,H,W", [(2, 640, 640, 32, 32), (1, 64, 96, 5, 8), (2, 1280, 1280, 16, 16), (3, 320, 320, 8, 8)])
def test_token_linear_into_nchw_residual(sd, oracle, gpu, rng, N, C, inner, H, W):
    """SpatialTransformer with Linear projections, the way out (SDXL, block.hpp:566-572): proj_out Linear (+bias) on tokens -> PERMUTE(1,0,2,3) -> CONT -> RESHAPE
    [W,H,C,N] -> ADD(., x_in).  Runs as ONE 1x1 implicit-GEMM conv over the token rows with the NCHW + bias + residual epilogue: no transposing copy, no add."""
    t = rng.standard_normal((N, H * W, inner)).astype(np.float32)
    xin = rng.standard_normal((N, C, H, W)).astype(np.float32)
    wl = (rng.standard_normal((C, inner)) / np.sqrt(inner)).astype(np.float32)
    bl = rng.standard_normal(C).astype(np.float32)

    def build(g, L):
        y = L.ggml_mul_mat(g.ctx, g.weight(wl, F16), g.input(t))
        y = L.ggml_add_inplace(g.ctx, y, g.weight(bl, F32))
        y = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, y, 1, 0, 2, 3))
        y = L.ggml_reshape_4d(g.ctx, y, W, H, C, N)
        return L.ggml_add(g.ctx, y, g.input(xin))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == (N, C, H, W) and np.isfinite(out).all()
    assert rel_l2(out, ref) < 3e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        assert sd.backend_stats()["fused_proj_tokens"] - before["fused_proj_tokens"] == 1


@pytest.mark.parametrize("N,Ca,Cb,HW,mean", [(2, 64, 32, 16, 1.5), (2, 320, 320, 64, -0.5), (1, 640, 320, 32, 0.2), (3, 128, 64, 8, 0.0)])
def test_skip_concat_group_norm_two_sources(sd, oracle, gpu, rng, N, Ca, Cb, HW, mean):
    """UNet skip connection (unet.hpp:702 + block.hpp:126-179): h = CONCAT(h, skip; channels); ResBlock(h) = conv3x3(SiLU(GN(h) w + b)) ... + conv1x1(h).  The
    concatenation is never built: GroupNorm statistics and ONE transposing pass read the two sources and write both convs' f16 NHWC operands (plan_concat_gn).
    Ca = 64 with 3 channels per group and Ca = 640 with 30: groups straddle the two sources; the sources carry different means."""
    C = Ca + Cb
    a = (rng.standard_normal((N, Ca, HW, HW)) * 1.3 + mean).astype(np.float32)
    b = (rng.standard_normal((N, Cb, HW, HW)) * 0.7 - mean).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    bb = rng.standard_normal(C).astype(np.float32)
    OC = 64
    wc = (rng.standard_normal((OC, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    ws = (rng.standard_normal((OC, C, 1, 1)) / np.sqrt(C)).astype(np.float32)

    def build(g, L):
        h = L.ggml_concat(g.ctx, g.input(a), g.input(b), 2)
        t = L.ggml_group_norm(g.ctx, h, 32, 1e-6)
        t = L.ggml_mul_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(w, F32), 1, 1, C, 1))
        t = L.ggml_add_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(bb, F32), 1, 1, C, 1))
        t = L.ggml_silu_inplace(g.ctx, t)
        y = L.ggml_conv_2d(g.ctx, g.weight(wc, F16), t, 1, 1, 1, 1, 1, 1)
        sk = L.ggml_conv_2d(g.ctx, g.weight(ws, F16), h, 1, 1, 0, 0, 1, 1)
        return L.ggml_add(g.ctx, y, sk)

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all() and rel_l2(out, ref) < 3e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        assert sd.backend_stats()["fused_concat_gn"] - before["fused_concat_gn"] == 1


@pytest.mark.parametrize("C,rows", [(320, 64), (1280, 17), (77, 5), (3072, 8)])
def test_layer_norm_chain(sd, oracle, gpu, rng, C, rows):
    x = (rng.standard_normal((rows, C)) * 2 + 1).astype(np.float32)
    w = rng.standard_normal(C).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)

    def build(g, L):
        t = L.ggml_norm(g.ctx, g.input(x), 1e-5)
        t = L.ggml_mul_inplace(g.ctx, t, g.weight(w, F32))
        return L.ggml_add_inplace(g.ctx, t, g.weight(b, F32))

    ref, out = run_both(sd, oracle, gpu, build)
    assert np.abs(out - ref).max() < 3e-5 * max(1.0, np.abs(ref).max())
    ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_rms_norm(g.ctx, g.input(x), 1e-6))
    assert np.abs(out - ref).max() < 3e-5


@pytest.mark.parametrize("C,rows,rms", [(320, 4101, False), (640, 16384, False), (1280, 4096, False), (300, 5000, False), (64, 4200, True), (1280, 4099, True)])
def test_layer_norm_into_linear_operand_image_multi_row_waves(sd, oracle, gpu, rng, C, rows, rms):
    """Round 5 (gemm16.hip k_layer_norm_f16_rows, option ln16_rows): LayerNorm / RMSNorm (+ affine) feeding a Linear writes the GEMM's f16 operand image; with >= 4096
    rows of <= 1280 values a wave handles 4 (C <= 512) or 2 rows with all loads in flight before the first reduction.  Per-row arithmetic is the single-row
    kernel's: bit-identical to the default single-row kernel (ragged row counts included); also against the oracle.  Measured slower, so off by default."""
    x = (rng.standard_normal((rows, C)) * 2 + 1).astype(np.float32)
    w = rng.standard_normal(C).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    wl = (rng.standard_normal((96, C)) / np.sqrt(C)).astype(np.float32)

    def build(g, L):
        t = (L.ggml_rms_norm if rms else L.ggml_norm)(g.ctx, g.input(x), 1e-5)
        t = L.ggml_mul_inplace(g.ctx, t, g.weight(w, F32))
        if not rms:
            t = L.ggml_add_inplace(g.ctx, t, g.weight(b, F32))
        return L.ggml_mul_mat(g.ctx, g.weight(wl, F16), t)

    ref, out = run_both(sd, oracle, gpu, build)
    assert rel_l2(out, ref) < 3e-4
    if _on_gpu():
        try:
            sd.backend_set_option("ln16_rows", 4)   # the multi-row kernel (measured slower, off by default: profiles/r06e_*)
            with Graph(gpu) as g:
                multi = g.run(build(g, sd.lib()))
        finally:
            sd.backend_set_option("ln16_rows", 1)
        np.testing.assert_array_equal(out, multi)


def test_soft_max(sd, oracle, gpu, rng):
    x = (rng.standard_normal((3, 4, 50, 77)) * 4).astype(np.float32)
    ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_soft_max(g.ctx, g.input(x)))
    assert np.abs(out - ref).max() < 2e-6


@pytest.mark.parametrize("wtype,tol", [(F16, 2e-4), (F32, 2e-3), (BF16, 1e-2), (Q8_0, 1e-2), (Q4_0, 3e-2)])
@pytest.mark.parametrize("tokens,K,M", [(77, 768, 320), (256, 320, 1280), (1, 320, 1280), (300, 1280, 320), (130, 64, 64)])
def test_linear_weight_gemm(sd, oracle, gpu, rng, wtype, tol, tokens, K, M):
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)

    def build(g, L):
        y = L.ggml_mul_mat(g.ctx, g.weight(w, wtype), g.input(x))
        return L.ggml_add_inplace(g.ctx, y, g.weight(b, F32))

    ref, out = run_both(sd, oracle, gpu, build)
    # F32 weights: oracle keeps activations f32, the MFMA path rounds them to f16 (stated tolerance 2e-3)
    assert rel_l2(out, ref) < tol
    exact = x.astype(np.float64) @ dequant(w, wtype).astype(np.float64).T + b
    if _on_gpu() or wtype in (F16, F32, BF16):   # the oracle itself (q8_0-quantised activations) is outside this bar for q8_0 / q4_0 weights
        assert rel_l2(out.reshape(tokens, M), exact) < (2e-3 if _on_gpu() or wtype != BF16 else 1e-2)


@pytest.mark.parametrize("N,C,inner,H,W", [(2, 320, 320, 16, 16), (1, 64, 96, 5, 7), (3, 128, 64, 8, 8)])
def test_spatial_transformer_projections_as_token_gemms(sd, oracle, gpu, rng, N, C, inner, H, W):
    """SpatialTransformer shell with conv projections (SD1.x, block.hpp:548-577): proj_in conv1x1 -> PERMUTE(1,2,0,3) -> CONT -> tokens,
    [a Linear on the tokens], tokens -> CONT(PERMUTE(1,0,2,3)) -> RESHAPE -> proj_out conv1x1 -> + x.  Both 1x1 convs run as token GEMMs and
    neither transposing copy is executed."""
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w_in = (rng.standard_normal((inner, C, 1, 1)) / np.sqrt(C)).astype(np.float32)
    b_in = rng.standard_normal(inner).astype(np.float32)
    w_mid = (rng.standard_normal((inner, inner)) / np.sqrt(inner)).astype(np.float32)
    w_out = (rng.standard_normal((C, inner, 1, 1)) / np.sqrt(inner)).astype(np.float32)
    b_out = rng.standard_normal(C).astype(np.float32)

    def build(g, L):
        xin = g.input(x)
        h = L.ggml_conv_2d(g.ctx, g.weight(w_in, F16), xin, 1, 1, 0, 0, 1, 1)
        h = L.ggml_add_inplace(g.ctx, h, L.ggml_reshape_4d(g.ctx, g.weight(b_in, F32), 1, 1, inner, 1))
        t = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, h, 1, 2, 0, 3))          # [inner, W, H, N]
        t = L.ggml_reshape_3d(g.ctx, t, inner, W * H, N)
        t = L.ggml_mul_mat(g.ctx, g.weight(w_mid, F16), t)                      # stands in for the transformer blocks
        y = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, t, 1, 0, 2, 3))            # [W*H, inner, N]
        y = L.ggml_reshape_4d(g.ctx, y, W, H, inner, N)
        y = L.ggml_conv_2d(g.ctx, g.weight(w_out, F16), y, 1, 1, 0, 0, 1, 1)
        y = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b_out, F32), 1, 1, C, 1))
        return L.ggml_add(g.ctx, y, xin)

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == (N, C, H, W) and np.isfinite(out).all()
    assert rel_l2(out, ref) < 3e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        assert sd.backend_stats()["fused_proj_tokens"] - before["fused_proj_tokens"] == 2


@pytest.mark.parametrize("tokens,K,M,res,mode", [(3328, 2048, 4096, True, 2), (3328, 2048, 4096, False, 2), (4352, 3072, 3072, True, 2), (3100, 2560, 4096, False, 2),
                                                 (4352, 3072, 6144, True, 3), (4200, 2048, 9216, False, 3)])
def test_linear_stream_k(sd, oracle, gpu, rng, tokens, K, M, res, mode):
    """Stream-K (option streamk, gemm16.hip k_gemm16<..., SK>): Linears whose tile count leaves the last round of a one-workgroup-per-CU tile mostly
    empty (13 x 16 = 208 tiles of 256 x 256 on 256 CUs; 17 x 12 = 204 for the FLUX linear2 width; a ragged last row tile) run as one
    persistent workgroup per CU over equal (tile, K-tile) ranges, tiles cut by a range boundary are summed by their last-arriving part in part
    order.  Against the plain launch (same kernel, whole tiles: differs by f32 summation order in the cut tiles only), against the exact
    product of the f16-rounded operands, and bit-identical between runs."""
    if not _on_gpu():
        pytest.skip("planner option of the MI355X backend")
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((tokens, M)).astype(np.float32)

    def build(g, L):
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x)), g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(r)) if res else y

    def run_gpu():
        with Graph(gpu) as g:
            return g.run(build(g, sd.lib())).reshape(tokens, M)

    try:
        sd.backend_set_option("streamk", 0)
        plain = run_gpu()
        sd.backend_set_option("streamk", mode)  # 2 = every candidate shape cut over all tiles; 3 = hybrid: whole tiles for the full rounds (17 x 24 = 408 and 17 x 36 = 612
        #                                          tiles of 256 x 256: one / two rounds of 256), the rest cut over K among all workgroups and computed first
        before = sd.backend_stats()["split_k_inlaunch"]
        out = run_gpu()
        assert sd.backend_stats()["split_k_inlaunch"] - before == 1, "the shape did not take the stream-K launch"
        again = run_gpu()
    finally:
        sd.backend_set_option("streamk", 0)
    assert np.isfinite(out).all()
    np.testing.assert_array_equal(out, again)
    assert rel_l2(out, plain) < 1e-6
    rows = np.unique(np.concatenate([rng.integers(0, tokens, 48), [0, 255, 256, tokens - 1]]))
    exact = x[rows].astype(np.float16).astype(np.float64) @ w.astype(np.float16).astype(np.float64).T + b + (r[rows] if res else 0.0)
    assert np.abs(out[rows] - exact).max() < 1e-3 * max(1.0, float(np.abs(exact).max()))


@pytest.mark.parametrize("tokens,K,M,res", [(4352, 2048, 12288, True), (8192, 2048, 2432, True), (4096, 2048, 7296, False), (8500, 2304, 9728, False), (4300, 2048, 9216, True)])
def test_linear_row_split_and_padded_256_tiles(sd, oracle, gpu, rng, tokens, K, M, res):
    """Round 5 tile policy of the DiT Linears (gemm16.hip g16_tail_rows / g16_pad256_ok):
      tail_split  17 x 48 = 816 tiles of 256 x 256 are 3.19 rounds on 256 CUs, paid as 4 — the launch runs as 768 whole tiles (rows 0..4095) plus the last 256 rows on
                  small tiles (second launch, row_base); no slab, no reduction;
      t256p_pad   widths that are multiples of 128 only (SD3.5: 2432, 7296) take the pipelined 256 x 256 tile, the last column tile half empty.
    Against the same Linear with both options off (same products, tile geometry only), against the exact product of the f16-rounded operands on sampled rows
    (first / last row of the main part, first row of the tail, last row; columns of the half-empty tile included), bit-identical between runs."""
    if not _on_gpu():
        pytest.skip("tile policy of the MI355X backend")
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((tokens, M)).astype(np.float32)

    def build(g, L):
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x)), g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(r)) if res else y

    def run_gpu():
        with Graph(gpu) as g:
            return g.run(build(g, sd.lib())).reshape(tokens, M)

    try:
        sd.backend_set_option("tail_split", 0)
        sd.backend_set_option("t256p_pad", 0)
        plain = run_gpu()
        sd.backend_set_option("tail_split", 1)
        sd.backend_set_option("t256p_pad", 1)
        out = run_gpu()
        again = run_gpu()
    finally:
        sd.backend_set_option("tail_split", 0)   # the library defaults (tail_split measured neutral: off; t256p_pad: on)
        sd.backend_set_option("t256p_pad", 1)
    assert np.isfinite(out).all()
    np.testing.assert_array_equal(out, again)
    assert rel_l2(out, plain) < 1e-6
    edges = [0, 255, 256, tokens - 1] + [k * 256 + d for k in range(1, (tokens + 255) // 256) for d in (-1, 0)]
    rows = np.unique(np.concatenate([rng.integers(0, tokens, 48), [e for e in edges if 0 <= e < tokens]]))
    exact = x[rows].astype(np.float16).astype(np.float64) @ w.astype(np.float16).astype(np.float64).T + b + (r[rows] if res else 0.0)
    assert np.abs(out[rows] - exact).max() < 1e-3 * max(1.0, float(np.abs(exact).max()))


@pytest.mark.parametrize("qname", ["Q8_0", "Q4_0"])
@pytest.mark.parametrize("tokens,K,M", [(640, 512, 384), (1030, 256, 200), (2048, 3072, 640)])
def test_quantised_linear_just_in_time_image(sd, oracle, gpu, rng, qname, tokens, K, M):
    """Resident-quantised mode (option jit_qimages): above the raw-block kernels' row range a q8_0 / q4_0 Linear keeps no f16 weight image — k_wswz_q
    rebuilds it from the raw GGUF blocks into a shared buffer in front of the GEMM.  The rebuilt image must equal the cached one bit for bit
    (same f16(d * q) values), so the two modes give IDENTICAL outputs; M = 200 exercises the zero rows that pad the image to 128."""
    if not _on_gpu():
        pytest.skip("planner option of the MI355X backend")
    wtype = Q8_0 if qname == "Q8_0" else Q4_0
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)

    def build(g, L):
        y = L.ggml_mul_mat(g.ctx, g.weight(w, wtype), g.input(x))
        return L.ggml_add_inplace(g.ctx, y, g.weight(b, F32))

    outs = []
    for jit in (0, 1):
        sd.backend_set_option("jit_qimages", jit)
        before = sd.backend_stats()["jit_images"]
        try:
            with Graph(gpu) as g:
                outs.append(g.run(build(g, sd.lib())))
        finally:
            sd.backend_set_option("jit_qimages", 4096)   # the default
        assert sd.backend_stats()["jit_images"] - before == jit
    assert np.array_equal(outs[0], outs[1])
    exact = x.astype(np.float16).astype(np.float64) @ dequant(w, wtype).astype(np.float64).T + b
    assert rel_l2(outs[1].reshape(tokens, M), exact) < 2e-3


@pytest.mark.parametrize("qname", ["Q8_0", "Q4_0"])
@pytest.mark.parametrize("tokens,K,M,epi", [(4096, 2048, 3072, "bias"), (4352, 3072, 3072, "res"), (4000, 2304, 3072, "gelu"), (6200, 2048, 2048, "bias"),
                                            (4096, 2048, 3072, "gate")])
def test_quantised_linear_dequantised_in_the_gemm_main_loop(sd, oracle, gpu, rng, qname, tokens, K, M, epi):
    """In-loop dequantisation (k_gemm16<..., QT>, option qinloop_min_rows): above k_qgemm16's row range a q8_0 / q4_0 Linear whose launch takes the pipelined
    256 x 256 tile reads the RAW GGUF blocks — LDS-DMA into a raw ring, dequantised once per workgroup into the B stage of the MFMA loop — and keeps no f16
    weight image, cached or rebuilt.  Same operand values (f16(d * q)) and the same summation order as the image path on the same tile: the two outputs must be
    IDENTICAL; ragged last row tiles (4000, 6200, 4352 rows), K stages 64 / 72 / 96, epilogues bias / residual / GELU -> f16 operand image / DiT gate."""
    if not _on_gpu():
        pytest.skip("planner option of the MI355X backend")
    wtype = Q8_0 if qname == "Q8_0" else Q4_0
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((tokens, M)).astype(np.float32)
    gt = rng.standard_normal((1, M)).astype(np.float32)
    w2 = (rng.standard_normal((256, M)) / np.sqrt(M)).astype(np.float32)

    def build(g, L):
        xin0 = g.input(x.reshape(1, tokens, K)) if epi == "gate" else g.input(x)
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), xin0), g.weight(b, F32))
        if epi == "res":
            y = L.ggml_add(g.ctx, y, g.input(r))
        elif epi == "gelu":   # fc1 -> GELU -> fc2 (f16 weights): fc1's epilogue writes fc2's operand image
            y = L.ggml_mul_mat(g.ctx, g.weight(w2, F16), L.ggml_gelu_inplace(g.ctx, y))
        elif epi == "gate":   # x + (Linear) * gate
            y = L.ggml_add(g.ctx, g.input(r.reshape(1, tokens, M)), L.ggml_mul(g.ctx, y, g.input(gt.reshape(1, 1, M))))
        return y

    outs = []
    for qin in (0, 513):
        sd.backend_set_option("qinloop_min_rows", qin)
        before = sd.backend_stats()["qinloop_linears"]
        try:
            with Graph(gpu) as g:
                outs.append(g.run(build(g, sd.lib())))
            with Graph(gpu) as g:
                again = g.run(build(g, sd.lib()))
        finally:
            sd.backend_set_option("qinloop_min_rows", 513)   # the default
        assert sd.backend_stats()["qinloop_linears"] - before == (2 if qin else 0)
        assert np.array_equal(outs[-1], again)
    assert np.isfinite(outs[1]).all()
    assert np.array_equal(outs[0], outs[1])
    rows = np.unique(np.concatenate([rng.integers(0, tokens, 40), [0, 255, 256, tokens - 1]]))
    lin = x[rows].astype(np.float16).astype(np.float64) @ dequant(w, wtype).astype(np.float64).T + b
    if epi == "res":
        exact = lin + r[rows]
    elif epi == "gate":
        exact = r[rows] + lin * gt
    elif epi == "gelu":
        gl = 0.5 * lin * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (lin + 0.044715 * lin ** 3)))
        exact = gl.astype(np.float16).astype(np.float64) @ w2.astype(np.float16).astype(np.float64).T
    else:
        exact = lin
    got = outs[1].reshape(tokens, -1)[rows]
    assert np.abs(got - exact).max() < 3e-3 * max(1.0, float(np.abs(exact).max()))


@pytest.mark.parametrize("d,H,L,M,flash", [(64, 2, 200, 384, True), (128, 1, 77, 128, True), (64, 3, 130, 320, False), (40, 2, 96, 112, True)])
def test_single_block_tail_concat_assembled_as_operand_image(sd, oracle, gpu, rng, d, H, L, M, flash):
    """FLUX single block tail (flux.hpp:594-700): t = linear1(x) -> q / k / v per-head views + mlp view; attn = flash(q, k, v) -> VIEW -> CONT;
    out = linear2(concat(attn, gelu(CONT(mlp view)), 0)).  The f32 concatenation is never built: linear2's f16 operand image [L][C + M] is
    filled by the flash kernel (columns 0 .. C) and by one strided-read -> GELU -> f16 pass (columns C ..).  flash = False feeds a plain f32
    tensor as the first part (packed at the CONCAT node); d = 40, M = 112: (C + M) % 64 != 0 keeps the unfused path."""
    C = d * H
    x = rng.standard_normal((1, L, 96)).astype(np.float32)
    w1 = (rng.standard_normal((3 * C + M, 96)) / np.sqrt(96)).astype(np.float32)
    b1 = (rng.standard_normal(3 * C + M) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((C, C + M)) / np.sqrt(C + M)).astype(np.float32)
    b2 = rng.standard_normal(C).astype(np.float32)
    a_in = rng.standard_normal((1, L, C)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)

    def build(g, L_):
        t = L_.ggml_mul_mat(g.ctx, g.weight(w1, F16), g.input(x))
        t = L_.ggml_add_inplace(g.ctx, t, g.weight(b1, F32))                       # [3C + M, L, 1]
        nb = sd_tensor_nb(t)
        if flash:
            def part(i, f16):
                p = L_.ggml_view_4d(g.ctx, t, d, H, L, 1, 4 * d, nb[1], nb[2], 4 * C * i)
                p = L_.ggml_cont(g.ctx, L_.ggml_permute(g.ctx, p, 0, 2, 1, 3))      # [d, L, H, 1]
                p = L_.ggml_reshape_3d(g.ctx, p, d, L, H)
                return L_.ggml_cast(g.ctx, p, F16) if f16 else p
            a = L_.ggml_flash_attn_ext(g.ctx, part(0, False), part(1, True), part(2, True), None, scale, 0.0, 0.0)
            L_.ggml_flash_attn_ext_set_prec(a, 10)
            na = sd_tensor_nb(a)
            a = L_.ggml_view_4d(g.ctx, a, d, H, L, 1, na[1], na[2], na[1] * H, 0)
            a = L_.ggml_cont(g.ctx, L_.ggml_permute(g.ctx, a, 0, 1, 2, 3))
            a = L_.ggml_reshape_3d(g.ctx, a, C, L, 1)
        else:
            a = g.input(a_in)
        m = L_.ggml_view_3d(g.ctx, t, M, L, 1, nb[1], nb[2], 4 * 3 * C)
        m = L_.ggml_gelu_inplace(g.ctx, L_.ggml_cont(g.ctx, m))
        y = L_.ggml_mul_mat(g.ctx, g.weight(w2, F16), L_.ggml_concat(g.ctx, a, m, 0))
        return L_.ggml_add_inplace(g.ctx, y, g.weight(b2, F32))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2      # the oracle's flash path accumulates V in f16, its GELU goes through the f16 table
    # exact chain in float64 on the f16-rounded operands at the reference's rounding points
    f16 = lambda v: np.asarray(v, np.float32).astype(np.float16).astype(np.float64)
    t = f16(x[0]) @ f16(w1).T + b1
    if flash:
        q, k, v = (t[:, i * C:(i + 1) * C].reshape(L, H, d).transpose(1, 0, 2) for i in range(3))
        att = _attn_exact(q, f16(k), f16(v), scale).transpose(1, 0, 2).reshape(L, C)
    else:
        att = a_in[0].astype(np.float64)
    u = t[:, 3 * C:]
    gel = 0.5 * u * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (u + 0.044715 * u ** 3)))
    exact = f16(np.concatenate([att, gel], 1)) @ f16(w2).T + b2
    assert rel_l2(out.reshape(L, C), exact) < 2e-3
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        assert sd.backend_stats()["fused_cat_rows16"] - before["fused_cat_rows16"] == (1 if (C + M) % 64 == 0 else 0)


@pytest.mark.parametrize("N,C,K,H,W", [(2, 320, 1280, 16, 16), (3, 64, 256, 5, 7), (1, 128, 512, 9, 9), (2, 96, 384, 8, 8)])
def test_ff2_residual_written_as_proj_out_operand_rows(sd, oracle, gpu, rng, N, C, K, H, W):
    """Tail of a SpatialTransformer (block.hpp:560-577): FF2 Linear (+bias) + residual -> CONT(PERMUTE(1,0,2,3)) -> RESHAPE -> proj_out conv1x1
    (+bias) + x.  The FF2 result is read only by the 1x1 conv, which rounds its input to f16: the GEMM epilogue writes acc + bias + residual
    as the conv's f16 operand rows (no f32 tensor, no pack pass).  C = 96 (not a multiple of 64) keeps the f32 + pack path."""
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    t_in = rng.standard_normal((N, H * W, K)).astype(np.float32)
    t_res = rng.standard_normal((N, H * W, C)).astype(np.float32)
    w2 = (rng.standard_normal((C, K)) / np.sqrt(K)).astype(np.float32)
    b2 = rng.standard_normal(C).astype(np.float32)
    w_out = (rng.standard_normal((C, C, 1, 1)) / np.sqrt(C)).astype(np.float32)
    b_out = rng.standard_normal(C).astype(np.float32)

    def build(g, L):
        xin = g.input(x)
        res = g.input(t_res)
        t = L.ggml_mul_mat(g.ctx, g.weight(w2, F16), g.input(t_in))
        t = L.ggml_add_inplace(g.ctx, t, g.weight(b2, F32))
        t = L.ggml_add(g.ctx, t, res)
        y = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, t, 1, 0, 2, 3))
        y = L.ggml_reshape_4d(g.ctx, y, W, H, C, N)
        y = L.ggml_conv_2d(g.ctx, g.weight(w_out, F16), y, 1, 1, 0, 0, 1, 1)
        y = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b_out, F32), 1, 1, C, 1))
        return L.ggml_add(g.ctx, y, xin)

    before = sd.backend_stats() if _on_gpu() else None
    if _on_gpu():
        sd.backend_set_option("fuse_rows16", 1)   # off by default (measured slightly slower inside SD1.5); the path must stay correct
    try:
        ref, out = run_both(sd, oracle, gpu, build)
    finally:
        if _on_gpu():
            sd.backend_set_option("fuse_rows16", 0)
    assert out.shape == (N, C, H, W) and np.isfinite(out).all()
    assert rel_l2(out, ref) < 3e-4
    # exact: f16-rounded operands at the reference's rounding points (FF2 input and weight, then the conv's input and weight), f64 sums
    h = t_in.astype(np.float16).astype(np.float64) @ w2.astype(np.float16).astype(np.float64).T + b2 + t_res
    h16 = h.astype(np.float32).astype(np.float16).astype(np.float64)
    y = h16 @ w_out.reshape(C, C).astype(np.float16).astype(np.float64).T + b_out
    exact = y.reshape(N, H, W, C).transpose(0, 3, 1, 2) + x
    assert rel_l2(out, exact) < 3e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        assert st["fused_rows16"] - before["fused_rows16"] == (1 if C % 64 == 0 else 0)
        assert st["fused_proj_tokens"] - before["fused_proj_tokens"] == 1


@pytest.mark.parametrize("N,C,OC,HW,split", [(2, 64, 320, 16, False), (3, 320, 96, 12, False), (2, 1280, 1280, 8, True), (2, 128, 128, 1, True), (1, 128, 64, 2, True),
                                              (5, 32, 32, 3, False)])
def test_conv_time_embedding_add_fused(sd, oracle, gpu, rng, N, C, OC, HW, split):
    """ResBlock (block.hpp:126-179): conv3x3 (+bias) -> ADD(h, Linear(SiLU(emb)) reshaped [1,1,OC,N]).  The embedding branch sits BETWEEN the
    conv chain and the ADD in graph order (DFS), so the fused conv kernel is emitted at the ADD's position and adds the per-(image, channel)
    value in its epilogue (also through the split-K reduce pass)."""
    x = rng.standard_normal((N, C, HW, HW)).astype(np.float32)
    w = (rng.standard_normal((OC, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.standard_normal(OC).astype(np.float32)
    emb = rng.standard_normal((N, 128)).astype(np.float32)
    we = (rng.standard_normal((OC, 128)) / np.sqrt(128)).astype(np.float32)
    be = rng.standard_normal(OC).astype(np.float32)

    def build(g, L):
        h = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 1, 1, 1, 1)
        h = L.ggml_add_inplace(g.ctx, h, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))
        e = L.ggml_mul_mat(g.ctx, g.weight(we, F16), L.ggml_silu(g.ctx, g.input(emb)))
        e = L.ggml_add_inplace(g.ctx, e, g.weight(be, F32))
        return L.ggml_add(g.ctx, h, L.ggml_reshape_4d(g.ctx, e, 1, 1, OC, N))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == (N, OC, HW, HW) and np.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        assert st["fused_chan_add"] - before["fused_chan_add"] == 1
        if split:
            assert st["split_k_gemms"] - before["split_k_gemms"] >= 1


@pytest.mark.parametrize("wtype,tol", [(Q8_0, 1e-2), (Q4_0, 3e-2)])
@pytest.mark.parametrize("tokens,K,M,res", [(1, 3072, 18432 // 8, False), (2, 256, 100, True), (2, 1024, 33, False), (1, 4096, 640, True), (1, 768, 96, False), (3, 768, 96, False),
                                            (4, 3072, 300, True), (5, 1024, 100, False), (8, 4096, 130, False), (9, 2048, 64, True), (16, 3072, 256, False),
                                            (16, 4096, 66, True), (16, 8192, 64, False), (17, 768, 96, False)])
@pytest.mark.parametrize("QGEMV_ROWS", [4, 16])   # 4 = the default policy, 16 = the kernel's range (option qgemv_max_rows)
def test_quantised_gemv_raw_blocks(sd, oracle, gpu, rng, wtype, tol, tokens, K, M, res, QGEMV_ROWS):
    """q8_0 / q4_0 Linear under 1 .. 16 activation rows (DiT adaLN / modulation vectors and embedders of a batch, ResBlock embedding projections):
    k_qgemv (1-2 rows) / k_qgemv_rows (3-16 rows, rows staged in LDS) stream the RAW GGUF blocks and dequantise in registers — no f16 weight
    image is built.  Rounding points = the MFMA path's (f16 activations, exact d * q weights, f32 accumulation), so the bars are those of
    test_linear_weight_gemm: <= 2e-3 vs the exact dequantised product, 1e-2 / 3e-2 vs the oracle (which quantises the activations to q8_0
    like ggml-cpu).  17 rows, or 16 rows x 8192 (256 KB of staged rows): the MFMA GEMM."""
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((tokens, M)).astype(np.float32)

    def build(g, L):
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), g.input(x)), g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(r)) if res else y

    if QGEMV_ROWS != 4 and (tokens <= 4 or not _on_gpu()):
        pytest.skip("same path as the default policy")
    before = sd.backend_stats() if _on_gpu() else None
    if _on_gpu():
        sd.backend_set_option("qgemv_max_rows", QGEMV_ROWS)
    try:
        ref, out = run_both(sd, oracle, gpu, build)
    finally:
        if _on_gpu():
            sd.backend_set_option("qgemv_max_rows", 4)
    assert np.isfinite(out).all()
    assert rel_l2(out, ref) < tol
    exact = x.astype(np.float64) @ dequant(w, wtype).astype(np.float64).T + b + (r if res else 0)
    if _on_gpu():
        assert rel_l2(out.reshape(tokens, M), exact) < 2e-3
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        taken = st["qgemv_linears"] - before["qgemv_linears"]
        assert taken == (1 if tokens <= QGEMV_ROWS and (tokens <= 2 or (4 if tokens <= 4 else 8 if tokens <= 8 else 16) * K * 2 <= 128 * 1024) else 0)
        if taken:
            assert st["swizzled_weight_bytes"] == before["swizzled_weight_bytes"]   # no f16 image was built for this weight
            # same Linear through the MFMA GEMM (f16 weight image): the two kernels must agree far inside the quantisation bars
            sd.backend_set_option("qgemv", 0)
            try:
                with Graph(gpu) as g2:
                    alt = g2.run(build(g2, sd.lib()))
            finally:
                sd.backend_set_option("qgemv", 1)
            assert rel_l2(out, alt) < 5e-4


@pytest.fixture()
def qgemm16_on(sd):
    """k_qgemm16 up to 512 rows (the default; set explicitly so the test does not depend on it)"""
    if _on_gpu():
        sd.backend_set_option("qgemm16_max_rows", 512)
    yield
    if _on_gpu():
        sd.backend_set_option("qgemm16_max_rows", 512)


@pytest.mark.parametrize("wtype,tol", [(Q8_0, 1e-2), (Q4_0, 3e-2)])
@pytest.mark.parametrize("tokens,K,M,res", [(17, 768, 96, False), (33, 256, 100, True), (77, 768, 320, False), (64, 1024, 640, True), (130, 3072, 1152, True),
                                            (300, 1280, 320, False), (512, 4096, 200, False), (257, 12288, 128, True), (600, 768, 96, False)])
def test_quantised_mfma_gemm_raw_blocks(sd, oracle, gpu, rng, qgemm16_on, wtype, tol, tokens, K, M, res):
    """q8_0 / q4_0 Linear under 17 .. 512 activation rows (text-stream Linears of the DiTs, text encoders; option qgemm16_max_rows = 512, the default):
    k_qgemm16 streams the RAW GGUF blocks, dequantises them in registers into MFMA B fragments (f16(d * q), bit-identical to what the f16
    weight image would hold) and never builds that image.  Bars as for test_linear_weight_gemm; against the same Linear on the f16-image GEMM
    only the f32 summation order differs.  600 rows: above qgemm16_max_rows, stays on the image path."""
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((tokens, M)).astype(np.float32)

    def build(g, L):
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), g.input(x)), g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(r)) if res else y

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all()
    assert rel_l2(out, ref) < tol
    exact = x.astype(np.float64) @ dequant(w, wtype).astype(np.float64).T + b + (r if res else 0)
    if _on_gpu():
        assert rel_l2(out.reshape(tokens, M), exact) < 2e-3
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        taken = st["qgemm16_linears"] - before["qgemm16_linears"]
        assert taken == (1 if tokens <= 512 else 0)
        if taken:
            assert st["swizzled_weight_bytes"] == before["swizzled_weight_bytes"]   # no f16 image was built for this weight
            sd.backend_set_option("qgemm16", 0)
            try:
                with Graph(gpu) as g2:
                    alt = g2.run(build(g2, sd.lib()))
            finally:
                sd.backend_set_option("qgemm16", 1)
            assert sd.backend_stats()["swizzled_weight_bytes"] > st["swizzled_weight_bytes"]
            assert rel_l2(out, alt) < 2e-5   # same f16 weights, same f16 activations: f32 summation order only


@pytest.mark.parametrize("wtype", [F16, F32])
@pytest.mark.parametrize("rows,K,M,res,silu", [(16, 1280, 320, False, True), (16, 1280, 1280, False, True), (16, 320, 1280, False, False), (1, 320, 1280, False, False),
                                                (5, 768, 100, True, False), (2, 2816, 1280, False, True), (9, 1280, 644, True, True), (16, 24, 8, False, True),
                                                (17, 1280, 320, False, True)])
def test_few_row_linear_weight_stream(sd, oracle, gpu, rng, wtype, rows, K, M, res, silu):
    """Linear with f16 / f32 weights under <= 16 activation rows (time_embed, ResBlock emb_layers: SiLU(emb) -> Linear, block.hpp:126-160): ONE
    k_fgemv launch; the SiLU node in front of it is not executed (applied while the rows are staged).  f16 weights: the MFMA path's rounding
    points (summation order only vs the GEMM); f32 weights: f32 x f32 like the oracle (the GEMM image rounds both to f16).  17 rows: GEMM."""
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((rows, M)).astype(np.float32)

    def build(g, L):
        h = g.input(x)
        if silu:
            h = L.ggml_silu(g.ctx, h)
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), h), g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(r)) if res else y

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all()
    taken = rows <= 16
    on_stream = _on_gpu() and not os.environ.get("SDCPP_BACKEND_OPTS")
    assert rel_l2(out, ref) < (2e-4 if wtype == F16 else (1e-5 if taken and on_stream else 2e-3))
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        # (silu + residual: the allocator may place the result on the rows the deferred SiLU would still have to read; the planner then runs
        # the SiLU and the GEMM path — correct, just not the streaming kernel)
        took = st["fgemv_linears"] - before["fgemv_linears"]
        assert took == (1 if taken else 0) or (taken and silu and res and took == 0)
        assert st["fused_presilu"] - before["fused_presilu"] == (1 if took and silu else 0)
        taken = bool(took)
        if taken:
            assert st["swizzled_weight_bytes"] == before["swizzled_weight_bytes"]   # no weight image, no pack, no split-K pass
            # one launch; with silu + residual the allocator may put the result on the (never written) SiLU buffer, which the residual fusion
            # reads as an operand overlap: the ADD then runs as its own kernel
            assert st["kernels_planned"] - before["kernels_planned"] <= (2 if silu and res else 1)
            sd.backend_set_option("fgemv", 0)
            try:
                with Graph(gpu) as g2:
                    alt = g2.run(build(g2, sd.lib()))
            finally:
                sd.backend_set_option("fgemv", 1)
            assert rel_l2(out, alt) < (2e-5 if wtype == F16 else 2e-3)


@pytest.mark.parametrize("wtype", [Q8_0, Q4_0])
def test_quantised_gemv_with_deferred_silu(sd, oracle, gpu, rng, wtype):
    """adaLN modulation with quantised weights (mmdit.hpp / flux.hpp: Linear(SiLU(vec)) on one row per image): the SiLU is applied by k_qgemv
    while it stages the rows."""
    rows, K, M = 2, 1024, 600
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)

    def build(g, L):
        return L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), L.ggml_silu(g.ctx, g.input(x))), g.weight(b, F32))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert rel_l2(out, ref) < (1e-2 if wtype == Q8_0 else 3e-2)
    xs = x / (1 + np.exp(-x.astype(np.float64)))
    exact = xs @ dequant(w, wtype).astype(np.float64).T + b
    if _on_gpu():
        assert rel_l2(out.reshape(rows, M), exact) < 2e-3
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        assert st["qgemv_linears"] - before["qgemv_linears"] == 1 and st["fused_presilu"] - before["fused_presilu"] == 1
        assert st["kernels_planned"] - before["kernels_planned"] == 1


@pytest.mark.parametrize("wtype", [Q8_0, Q4_0])
def test_quantised_mlp_gelu_chain_raw_blocks(sd, oracle, gpu, rng, qgemm16_on, wtype):
    """Mlp (block.hpp:249-258) with quantised weights on a short token run: fc1 -> GELU is written by k_qgemm16 as the f16 operand image of
    fc2, fc2 (+bias, +residual) reads it — both Linears on raw blocks."""
    tokens, C, Hd = 154, 256, 1024
    x = rng.standard_normal((tokens, C)).astype(np.float32)
    w1 = (rng.standard_normal((Hd, C)) / np.sqrt(C)).astype(np.float32)
    b1 = rng.standard_normal(Hd).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((C, Hd)) / np.sqrt(Hd)).astype(np.float32)
    b2 = rng.standard_normal(C).astype(np.float32) * 0.1

    def build(g, L):
        xin = g.input(x)
        h = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w1, wtype), xin), g.weight(b1, F32))
        h = L.ggml_gelu_inplace(g.ctx, h)
        y = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w2, wtype), h), g.weight(b2, F32))
        return L.ggml_add(g.ctx, y, xin)

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all()
    assert rel_l2(out, ref) < (1e-2 if wtype == Q8_0 else 3e-2)
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        assert st["qgemm16_linears"] - before["qgemm16_linears"] == 2
        assert st["fused_gelu"] - before["fused_gelu"] == 1
        assert st["swizzled_weight_bytes"] == before["swizzled_weight_bytes"]
        sd.backend_set_option("qgemm16", 0)
        try:
            with Graph(gpu) as g2:
                alt = g2.run(build(g2, sd.lib()))
        finally:
            sd.backend_set_option("qgemm16", 1)
        assert rel_l2(out, alt) < 1e-4   # the f16 rounding of gelu(fc1) can flip on summation-order differences


def test_linear_residual_fusion_and_batch_dims(sd, oracle, gpu, rng):
    # [C, L, N] activations, bias + residual: the BasicTransformerBlock tail (block.hpp:450-466)
    x = rng.standard_normal((2, 64, 320)).astype(np.float32)
    r = rng.standard_normal((2, 64, 640)).astype(np.float32)
    w = (rng.standard_normal((640, 320)) / 18).astype(np.float32)
    b = rng.standard_normal(640).astype(np.float32)

    def build(g, L):
        y = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))
        y = L.ggml_add_inplace(g.ctx, y, g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(r))

    ref, out = run_both(sd, oracle, gpu, build)
    assert rel_l2(out, ref) < 2e-4


def test_generic_matmul_batched(sd, oracle, gpu, rng):
    # activations x activations (manual attention scores): exact f32 on both sides
    k = rng.standard_normal((6, 77, 40)).astype(np.float32)
    q = rng.standard_normal((6, 100, 40)).astype(np.float32)
    ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_mul_mat(g.ctx, g.input(k), g.input(q)))
    assert rel_l2(out, ref) < 1e-5


@pytest.mark.parametrize("N,IC,OC,H,W,ks,stride,pad", [
    (2, 4, 320, 16, 16, 3, 1, 1),      # UNet input conv (IC=4 padded to 32)
    (1, 320, 320, 32, 32, 3, 1, 1),    # ResBlock conv
    (2, 64, 96, 16, 16, 3, 2, 1),      # downsample
    (1, 320, 4, 16, 16, 3, 1, 1),      # out conv (OC=4 padded to 64)
    (2, 96, 64, 8, 8, 1, 1, 0),        # 1x1 proj / skip
    (1, 32, 48, 12, 20, 3, 1, 1),      # non-power-of-two map (masked tiles)
    (1, 128, 128, 128, 128, 3, 1, 1),  # VAE-sized map (wide tile)
])
def test_conv2d_chain(sd, oracle, gpu, rng, N, IC, OC, H, W, ks, stride, pad):
    x = rng.standard_normal((N, IC, H, W)).astype(np.float32)
    w = (rng.standard_normal((OC, IC, ks, ks)) / np.sqrt(IC * ks * ks)).astype(np.float32)
    b = rng.standard_normal(OC).astype(np.float32)

    def build(g, L):
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), stride, stride, pad, pad, 1, 1)
        return L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))

    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-4


def test_conv2d_direct_and_residual(sd, oracle, gpu, rng):
    x = rng.standard_normal((2, 64, 16, 16)).astype(np.float32)
    w = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_conv_2d_direct(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 1, 1, 1, 1))
    assert rel_l2(out, ref) < 2e-4

    # ResBlock tail: GN -> SiLU -> conv -> +bias -> +x   (the allocator may recycle the conv input for the output)
    gw = rng.standard_normal(64).astype(np.float32)
    gb = rng.standard_normal(64).astype(np.float32)

    def build(g, L):
        xin = g.input(x)
        t = L.ggml_group_norm(g.ctx, xin, 32, 1e-6)
        t = L.ggml_mul_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(gw, F32), 1, 1, 64, 1))
        t = L.ggml_add_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(gb, F32), 1, 1, 64, 1))
        t = L.ggml_silu_inplace(g.ctx, t)
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), t, 1, 1, 1, 1, 1, 1)
        y = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, 64, 1))
        return L.ggml_add(g.ctx, y, xin)

    ref, out = run_both(sd, oracle, gpu, build)
    assert rel_l2(out, ref) < 2e-4


def test_layout_ops(sd, oracle, gpu, rng):
    x = rng.standard_normal((2, 24, 10, 12)).astype(np.float32)   # [N,C,H,W]
    y = rng.standard_normal((2, 8, 10, 12)).astype(np.float32)

    def nchw_to_tokens(g, L):
        t = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, g.input(x), 1, 2, 0, 3))
        return L.ggml_reshape_3d(g.ctx, t, 24, 120, 2)

    def tokens_to_nchw(g, L):
        t = L.ggml_reshape_3d(g.ctx, g.input(x), 12 * 10, 24, 2)     # treat as [HW? ...] generic 2-D transpose
        return L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, t, 1, 0, 2, 3))

    def head_split(g, L):
        t = L.ggml_reshape_4d(g.ctx, g.input(x), 12, 10, 24, 2)
        return L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, t, 0, 2, 1, 3))

    for build in (nchw_to_tokens, tokens_to_nchw, head_split,
                  lambda g, L: L.ggml_concat(g.ctx, g.input(x), g.input(y), 2),
                  lambda g, L: L.ggml_upscale(g.ctx, g.input(x), 2, 0),
                  lambda g, L: L.ggml_cast(g.ctx, g.input(x), F16),
                  lambda g, L: L.ggml_repeat(g.ctx, g.input(x[:1]), g.input(x)),
                  lambda g, L: L.ggml_pad(g.ctx, g.input(x), 1, 1, 0, 0)):
        ref, out = run_both(sd, oracle, gpu, build)
        assert out.shape == ref.shape
        np.testing.assert_array_equal(out, ref)


def test_geglu_chain(sd, oracle, gpu, rng):
    x = rng.standard_normal((2, 50, 64)).astype(np.float32)
    w = (rng.standard_normal((256, 64)) / 8).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)

    def build(g, L):
        h = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))
        h = L.ggml_add_inplace(g.ctx, h, g.weight(b, F32))
        ts = sd_tensor_nb(h)
        lo = L.ggml_view_4d(g.ctx, h, 128, 50, 2, 1, ts[1], ts[2], ts[3], 0)
        hi = L.ggml_view_4d(g.ctx, h, 128, 50, 2, 1, ts[1], ts[2], ts[3], 128 * 4)
        gate = L.ggml_gelu_inplace(g.ctx, L.ggml_cont(g.ctx, hi))
        return L.ggml_mul(g.ctx, lo, gate)

    ref, out = run_both(sd, oracle, gpu, build)
    assert rel_l2(out, ref) < 1e-3   # oracle GELU goes through the f16 table


def sd_tensor_nb(t):
    from ggml_graph import tensor_struct
    s = tensor_struct(t)
    return [int(s.nb[i]) for i in range(4)]


def _attn_exact(q, k, v, scale):
    s = np.einsum("hqd,hkd->hqk", q.astype(np.float64), k.astype(np.float64)) * scale
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return np.einsum("hqk,hkd->hqd", p, v.astype(np.float64))


@pytest.mark.parametrize("d,Lq,Lk,HN", [(40, 200, 200, 4), (40, 130, 77, 8), (64, 256, 256, 3), (80, 64, 77, 2), (160, 64, 64, 2), (16, 70, 70, 4),
                                        # d = 96 / 128 instantiations (FLUX: 24 heads x 128, L = 4096 + 256) with ragged key counts
                                        (96, 200, 333, 2), (96, 130, 4352, 1), (128, 333, 333, 3), (128, 512, 4352, 2), (128, 77, 77, 2)])
def test_flash_attn_ext(sd, oracle, gpu, rng, d, Lq, Lk, HN):
    """FLASH_ATTN_EXT node (f16 K/V).  The oracle reproduces ggml-cpu's F16 accumulation of V (Appendix E.3), which is
    LESS accurate than the MFMA kernel (f32 accumulation): tolerance vs oracle 1e-2 rel-L2, vs exact math 2e-3."""
    q = rng.standard_normal((HN, Lq, d)).astype(np.float32)
    k = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    v = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)

    def build(g, L):
        out = L.ggml_flash_attn_ext(g.ctx, g.input(q), g.input(k, F16), g.input(v, F16), None, scale, 0.0, 0.0)
        L.ggml_flash_attn_ext_set_prec(out, 10)
        return out

    with Graph(gpu) as g:
        node = build(g, sd.lib())
        assert g.supports(node)
    ref, out = run_both(sd, oracle, gpu, build)          # [1, Lq, HN, d]
    assert out.shape == ref.shape == (1, Lq, HN, d)
    # the oracle's F16 V accumulation loses ~sqrt(Lk) * 2^-11: over 4352 keys it sits ~1e-2 from the exact result by itself (the bar against
    # the exact softmax below is the one that pins the kernel; r02g: 1.01e-2 here on a passing kernel)
    assert rel_l2(out, ref) < (1e-2 if Lk <= 1024 else 3e-2)
    exact = _attn_exact(q, k.astype(np.float16).astype(np.float32), v.astype(np.float16).astype(np.float32), scale)  # [HN, Lq, d]
    assert rel_l2(out[0].transpose(1, 0, 2), exact) < (2e-3 if _on_gpu() else 1e-2)   # self-check mode: the oracle's f16 V accumulation


@pytest.mark.parametrize("d,H,Lq,Lk,N,ctx", [(40, 8, 300, 300, 2, 320), (40, 8, 130, 77, 3, 768), (80, 4, 256, 77, 1, 768), (160, 2, 64, 64, 2, 320), (64, 5, 96, 96, 1, 320)])
def test_attention_block_flash_operands_from_projections(sd, oracle, gpu, rng, d, H, Lq, Lk, N, ctx):
    """CrossAttention as the reference builds it with the flash flag on (ggml_extend.hpp:1349-1485): q/k/v Linears -> reshape / permute / cont
    (-> f16 cast for k, v) -> FLASH_ATTN_EXT -> view / cont -> to_out Linear.  The projections write the flash kernel's operand layouts
    directly: K/V as f16 head-major, and Q — read by nothing else — as an f16 head-major image too (stat fused_q16)."""
    C = d * H
    x = rng.standard_normal((N, Lq, C)).astype(np.float32)
    c = x if ctx == C and Lk == Lq else rng.standard_normal((N, Lk, ctx)).astype(np.float32)
    wq = (rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)
    wk = (rng.standard_normal((C, ctx)) / np.sqrt(ctx)).astype(np.float32)
    wv = (rng.standard_normal((C, ctx)) / np.sqrt(ctx)).astype(np.float32)
    wo = (rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)
    bo = rng.standard_normal(C).astype(np.float32)
    scale = 1.0 / np.sqrt(d)

    def build(g, L):
        xi = g.input(x)
        ci = xi if c is x else g.input(c)

        def heads(t, Lt, f16):
            t = L.ggml_reshape_4d(g.ctx, t, d, H, Lt, N)
            t = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, t, 0, 2, 1, 3))   # [d, L, H, N]
            t = L.ggml_reshape_3d(g.ctx, t, d, Lt, H * N)
            return L.ggml_cast(g.ctx, t, F16) if f16 else t

        q = heads(L.ggml_mul_mat(g.ctx, g.weight(wq, F16), xi), Lq, False)
        k = heads(L.ggml_mul_mat(g.ctx, g.weight(wk, F16), ci), Lk, True)
        v = heads(L.ggml_mul_mat(g.ctx, g.weight(wv, F16), ci), Lk, True)
        a = L.ggml_flash_attn_ext(g.ctx, q, k, v, None, scale, 0.0, 0.0)      # [d, H*N, Lq, 1]
        L.ggml_flash_attn_ext_set_prec(a, 10)
        nb = sd_tensor_nb(a)
        a = L.ggml_view_4d(g.ctx, a, d, H, Lq, N, nb[1], nb[2], nb[1] * H, 0)
        a = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, a, 0, 1, 2, 3))
        a = L.ggml_reshape_3d(g.ctx, a, C, Lq, N)
        y = L.ggml_mul_mat(g.ctx, g.weight(wo, F16), a)
        return L.ggml_add_inplace(g.ctx, y, g.weight(bo, F32))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == ref.shape == (1, N, Lq, C)
    out, ref = out[0], ref[0]
    assert rel_l2(out, ref) < 1e-2      # the oracle's flash path accumulates V in f16
    # exact chain in float64 on the f16-rounded weights
    f = lambda w: w.astype(np.float16).astype(np.float64)
    qe = (x.astype(np.float64) @ f(wq).T).reshape(N, Lq, H, d).transpose(0, 2, 1, 3).reshape(N * H, Lq, d)
    ke = (c.astype(np.float64) @ f(wk).T).reshape(N, Lk, H, d).transpose(0, 2, 1, 3).reshape(N * H, Lk, d)
    ve = (c.astype(np.float64) @ f(wv).T).reshape(N, Lk, H, d).transpose(0, 2, 1, 3).reshape(N * H, Lk, d)
    ae = _attn_exact(qe, ke, ve, scale).reshape(N, H, Lq, d).transpose(0, 2, 1, 3).reshape(N, Lq, C)
    exact = ae @ f(wo).T + bo
    assert rel_l2(out, exact) < (4e-3 if _on_gpu() else 1e-2)
    if before is not None and Lq >= 32:
        after = sd.backend_stats()
        assert after["fused_q16"] - before["fused_q16"] == (1 if d % 8 == 0 else 0)
        assert after["fused_attention"] - before["fused_attention"] == 1
        if not os.environ.get("SDCPP_BACKEND_OPTS"):
            # the projections reading the same activation run as ONE launch: q / k / v of a self-attention, k / v of a cross-attention
            assert after["fused_sibling_linears"] - before["fused_sibling_linears"] == (2 if c is x else 1)
            sd.backend_set_option("fuse_siblings", 0)
            try:
                with Graph(gpu) as g2:
                    alt = g2.run(build(g2, sd.lib()))
            finally:
                sd.backend_set_option("fuse_siblings", 1)
            assert sd.backend_stats()["fused_sibling_linears"] == after["fused_sibling_linears"]
            np.testing.assert_array_equal(out, alt[0])   # same k order per output element whatever the tile geometry: bit-identical


@pytest.mark.parametrize("d,H,Lq,Lk,N,ctx,blocks", [(40, 8, 96, 77, 2, 768, 3), (64, 4, 130, 77, 1, 320, 9), (80, 2, 64, 40, 3, 128, 2)])
def test_cross_attention_kv_of_all_blocks_hoisted(sd, oracle, gpu, rng, d, H, Lq, Lk, N, ctx, blocks):
    """`blocks` cross-attention layers in sequence reading ONE text context (what a UNet forward does: block.hpp CrossAttention to_k / to_v of every
    transformer block): all their K / V projections run as grouped multi-weight launches at the position of the first one, results kept in the
    backend's arena and read from there by the FLASH_ATTN_EXT nodes.  9 blocks = 18 projections = two launches (16 + 2)."""
    C = d * H
    x = rng.standard_normal((N, Lq, C)).astype(np.float32)
    c = rng.standard_normal((N, Lk, ctx)).astype(np.float32)
    ws = [[(rng.standard_normal(sh) / np.sqrt(sh[1])).astype(np.float32) for sh in ((C, C), (C, ctx), (C, ctx), (C, C))] for _ in range(blocks)]
    scale = 1.0 / np.sqrt(d)

    def build(g, L):
        h = g.input(x)
        ci = g.input(c)

        def heads(t, Lt, f16):
            t = L.ggml_reshape_4d(g.ctx, t, d, H, Lt, N)
            t = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, t, 0, 2, 1, 3))
            t = L.ggml_reshape_3d(g.ctx, t, d, Lt, H * N)
            return L.ggml_cast(g.ctx, t, F16) if f16 else t

        for wq, wk, wv, wo in ws:
            q = heads(L.ggml_mul_mat(g.ctx, g.weight(wq, F16), h), Lq, False)
            k = heads(L.ggml_mul_mat(g.ctx, g.weight(wk, F16), ci), Lk, True)
            v = heads(L.ggml_mul_mat(g.ctx, g.weight(wv, F16), ci), Lk, True)
            a = L.ggml_flash_attn_ext(g.ctx, q, k, v, None, scale, 0.0, 0.0)
            L.ggml_flash_attn_ext_set_prec(a, 10)
            nb = sd_tensor_nb(a)
            a = L.ggml_view_4d(g.ctx, a, d, H, Lq, N, nb[1], nb[2], nb[1] * H, 0)
            a = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, a, 0, 1, 2, 3))
            a = L.ggml_reshape_3d(g.ctx, a, C, Lq, N)
            h = L.ggml_add(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(wo, F16), a), h)
        return h

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all() and rel_l2(out, ref) < 1e-2      # the oracle's flash path accumulates V in f16
    if before is not None and Lk >= 32 and not os.environ.get("SDCPP_BACKEND_OPTS"):
        after = sd.backend_stats()
        assert after["hoisted_kv_linears"] - before["hoisted_kv_linears"] == (2 * blocks if blocks >= 2 and 2 * blocks >= 3 else 0)
        sd.backend_set_option("hoist_kv", 0)
        try:
            with Graph(gpu) as g2:
                alt = g2.run(build(g2, sd.lib()))
        finally:
            sd.backend_set_option("hoist_kv", 1)
        assert sd.backend_stats()["hoisted_kv_linears"] == after["hoisted_kv_linears"]
        np.testing.assert_array_equal(out, alt)


@pytest.mark.parametrize("d,Lq,Lk,HN", [(40, 200, 200, 4), (80, 100, 77, 2), (160, 64, 64, 2),
                                        # head dims beyond the flash kernel (KL-VAE mid attention: 1 head x 512): composed from MFMA GEMMs + f16 row softmax
                                        (512, 256, 256, 1), (192, 100, 80, 2), (512, 1024, 1024, 2)])
def test_manual_attention_chain(sd, oracle, gpu, rng, d, Lq, Lk, HN):
    """flash flag off: MUL_MAT(k,q) -> SCALE -> SOFT_MAX -> MUL_MAT(vT,kq) is routed to the flash kernel (f16 MFMA operands);
    the oracle computes this chain in exact f32, so the tolerance is the f16-operand one: 2e-3 rel-L2."""
    q = rng.standard_normal((HN, Lq, d)).astype(np.float32)
    k = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    vt = rng.standard_normal((HN, d, Lk)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)

    def build(g, L):
        kq = L.ggml_mul_mat(g.ctx, g.input(k), g.input(q))
        kq = L.ggml_scale_inplace(g.ctx, kq, scale)
        kq = L.ggml_soft_max_inplace(g.ctx, kq)
        return L.ggml_mul_mat(g.ctx, g.input(vt), kq)

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all() and rel_l2(out, ref) < (2e-3 if d <= 160 else 4e-3)   # d > 160: Q, K, P and V all enter the MFMA as f16
    if before is not None and d > 160 and not os.environ.get("SDCPP_BACKEND_OPTS"):
        assert sd.backend_stats()["gemm_attention"] - before["gemm_attention"] == 1


@pytest.mark.parametrize("tokens,K,M", [(24, 1280, 1280), (1000, 5120, 200), (130, 2560, 96)])   # (<= 16 rows: k_fgemv, no split)
def test_linear_split_k(sd, oracle, gpu, rng, tokens, K, M):
    """deep-K GEMMs over few output tiles run split-K (slabs + fixed-order reduce with bias + residual)"""
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((tokens, M)).astype(np.float32)

    def build(g, L):
        y = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))
        y = L.ggml_add_inplace(g.ctx, y, g.weight(b, F32))
        return L.ggml_add(g.ctx, y, g.input(r))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert rel_l2(out, ref) < 2e-4
    if before is not None:
        assert sd.backend_stats()["split_k_gemms"] == before["split_k_gemms"] + 1


@pytest.mark.parametrize("N,IC,OC,H,W,ks,stride", [(2, 256, 256, 8, 8, 3, 1), (1, 640, 320, 8, 8, 3, 1), (2, 128, 192, 16, 16, 3, 2), (1, 1280, 130, 6, 10, 1, 1)])
def test_conv2d_split_k(sd, oracle, gpu, rng, N, IC, OC, H, W, ks, stride):
    x = rng.standard_normal((N, IC, H, W)).astype(np.float32)
    w = (rng.standard_normal((OC, IC, ks, ks)) / np.sqrt(IC * ks * ks)).astype(np.float32)
    b = rng.standard_normal(OC).astype(np.float32)
    pad = ks // 2

    def build(g, L):
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), stride, stride, pad, pad, 1, 1)
        return L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-4
    if before is not None:
        assert sd.backend_stats()["split_k_gemms"] == before["split_k_gemms"] + 1


@pytest.mark.parametrize("N,IC,OC,HW", [(16, 1280, 1280, 16), (16, 1280, 1280, 8), (16, 2560, 1280, 16), (16, 2560, 1280, 8), (8, 1280, 1280, 16), (16, 1920, 640, 16)])
def test_conv2d_weight_major_workgroup_order(sd, oracle, gpu, rng, N, IC, OC, HW):
    """Round 5 (g16_common.h g16_wg_order, option conv_wmajor): the 8x8 / 16x16-level 3x3 convs of the benchmarked SD1.5 batch (16 images; 29.5 .. 59 MB of weights
    against 2.6 .. 21 MB of NHWC image, K cut into 4 .. 16 slices) run their workgroups weight-major — every (column tile, K slice) weight chunk on one XCD.  Every
    (tile, slice) computes what it computed before and the slab reduce sums in slice order, so the result is BIT-IDENTICAL to the default order; also against the
    oracle.  Both kernels that take such shapes: k_conv3w<16> (16-wide maps) and the implicit-GEMM k_gemm16<256, 320, true> (8-wide maps)."""
    x = rng.standard_normal((N, IC, HW, HW)).astype(np.float32)
    w = (rng.standard_normal((OC, IC, 3, 3)) / np.sqrt(IC * 9)).astype(np.float32)
    b = rng.standard_normal(OC).astype(np.float32)
    r = rng.standard_normal((N, OC, HW, HW)).astype(np.float32)

    def build(g, L):
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 1, 1, 1, 1)
        y = L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))
        return L.ggml_add(g.ctx, y, g.input(r))

    if not _on_gpu():
        pytest.skip("workgroup order of the MI355X kernels")
    with Graph(oracle) as g:
        ref = g.run(build(g, sd.lib()))

    def run_gpu():
        with Graph(gpu) as g:
            return g.run(build(g, sd.lib()))

    try:
        sd.backend_set_option("conv_wmajor", 0)
        plain = run_gpu()
        sd.backend_set_option("conv_wmajor", 1)
        out = run_gpu()
    finally:
        sd.backend_set_option("conv_wmajor", 1)
    assert np.isfinite(out).all()
    np.testing.assert_array_equal(out, plain)
    assert rel_l2(out, ref) < 2e-4


@pytest.mark.parametrize("N,C,H,W,res", [(8, 256, 8, 8, True), (8, 320, 16, 16, False), (8, 1280, 16, 16, True), (8, 640, 32, 32, True), (2, 256, 8, 8, True)])
def test_split_conv_reduce_writes_group_norm_statistics(sd, oracle, gpu, rng, N, C, H, W, res):
    """ResBlock seam (block.hpp:126-179 -> next block's norm): a split-K 3x3 conv (+bias, +residual) whose result is read by GroupNorm -> affine -> SiLU -> conv
    AND by a later ADD.  The slab reduce pass writes the values and the GroupNorm's per-(image, channel) affine (k_splitk_reduce_gn: register-resident
    groups of 512 / 2560 / 10240 / 20480 values), so both the conv values and the normalised branch are checked.  With fewer than 256 (image, group)
    pairs the planner keeps the separate statistics pass (last case)."""
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    r = (rng.standard_normal((N, C, H, W)) + 0.7).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    gw = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    gb = rng.standard_normal(C).astype(np.float32)
    w2 = (rng.standard_normal((C, C, 1, 1)) / np.sqrt(C)).astype(np.float32)

    def build(g, L):
        h = L.ggml_conv_2d(g.ctx, g.weight(w, F16), g.input(x), 1, 1, 1, 1, 1, 1)
        h = L.ggml_add_inplace(g.ctx, h, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, C, 1))
        if res:
            h = L.ggml_add(g.ctx, h, g.input(r))
        t = L.ggml_group_norm(g.ctx, h, 32, 1e-6)
        t = L.ggml_mul_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(gw, F32), 1, 1, C, 1))
        t = L.ggml_add_inplace(g.ctx, t, L.ggml_reshape_4d(g.ctx, g.weight(gb, F32), 1, 1, C, 1))
        t = L.ggml_silu_inplace(g.ctx, t)
        t = L.ggml_conv_2d(g.ctx, g.weight(w2, F16), t, 1, 1, 0, 0, 1, 1)
        return L.ggml_add(g.ctx, t, h)

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == (N, C, H, W) and np.isfinite(out).all()
    assert rel_l2(out, ref) < 3e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        assert st["split_k_gemms"] - before["split_k_gemms"] >= 1
        assert st["fused_gn_stats"] - before["fused_gn_stats"] == (1 if N * 32 >= 256 else 0)


@pytest.mark.parametrize("tokens,K,M,res", [(2048, 5120, 1280, True), (2048, 1280, 1280, True), (1000, 2560, 640, True), (2048, 1280, 320, False)])
def test_split_linear_reduce_writes_layer_norm_image(sd, oracle, gpu, rng, tokens, K, M, res):
    """Transformer-block seam (block.hpp BasicTransformerBlock: x = x + attn(norm1(x)); norm2(x) -> ...): a split-K Linear (+bias, +residual) whose result
    is read by LayerNorm -> affine feeding only a weight GEMM AND by a later ADD.  The slab reduce pass writes the f32 rows and the LayerNorm's f16
    operand image (k_splitk_reduce_ln); both branches are checked (SDXL's 2048-token level: to_out / FF2 -> the next LayerNorm)."""
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    r = (rng.standard_normal((tokens, M)) + 0.5).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal(M)).astype(np.float32)
    lb = rng.standard_normal(M).astype(np.float32)
    w2 = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float32)

    def build(g, L):
        h = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))
        h = L.ggml_add_inplace(g.ctx, h, g.weight(b, F32))
        if res:
            h = L.ggml_add(g.ctx, h, g.input(r))
        t = L.ggml_norm(g.ctx, h, 1e-5)
        t = L.ggml_mul_inplace(g.ctx, t, g.weight(lw, F32))
        t = L.ggml_add_inplace(g.ctx, t, g.weight(lb, F32))
        t = L.ggml_mul_mat(g.ctx, g.weight(w2, F16), t)
        return L.ggml_add(g.ctx, t, h)

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape[-2:] == (tokens, M) and np.isfinite(out).all()
    assert rel_l2(out, ref) < 3e-4
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        split = st["split_k_gemms"] - before["split_k_gemms"] + st["split_k_inlaunch"] - before["split_k_inlaunch"]
        # the fusion is taken exactly when the first Linear runs split-K with the slab reduce pass
        assert st["fused_ln_reduce"] - before["fused_ln_reduce"] in ((0, 1) if split else (0,))
        if tokens == 2048 and K >= 1280 and M == 1280:
            assert st["fused_ln_reduce"] - before["fused_ln_reduce"] == 1  # the SDXL shapes the fusion exists for


@pytest.mark.parametrize("d,H,L_,N,K,f16", [(40, 8, 77, 3, 768, True), (40, 8, 77, 3, 768, False), (80, 4, 200, 2, 320, True), (64, 2, 24, 5, 128, False),
                                            (160, 2, 64, 2, 320, True), (40, 2, 300, 1, 64, False)])
def test_projection_head_major_chain(sd, oracle, gpu, rng, d, H, L_, N, K, f16):
    """q/k/v projection + reshape/permute/cont (+ f16 cast): the GEMM epilogue writes the attention operand layout directly
    (ggml_extend.hpp:1349-1400); rows = N*L is ragged against the 128-row tile and L is not a multiple of 32."""
    C = d * H
    x = rng.standard_normal((N, L_, K)).astype(np.float32)
    w = (rng.standard_normal((C, K)) / np.sqrt(K)).astype(np.float32)

    def build(g, L):
        y = L.ggml_mul_mat(g.ctx, g.weight(w, F16), g.input(x))          # [C, L, N]
        y = L.ggml_reshape_4d(g.ctx, y, d, H, L_, N)
        y = L.ggml_cont(g.ctx, L.ggml_permute(g.ctx, y, 0, 2, 1, 3))     # [d, L, H, N]
        y = L.ggml_reshape_3d(g.ctx, y, d, L_, H * N)
        if f16:
            y = L.ggml_cast(g.ctx, y, F16)
            y = L.ggml_cast(g.ctx, y, F32)
        return y

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < (1e-3 if f16 else 2e-4)
    if before is not None:
        assert sd.backend_stats()["head_major_gemms"] == before["head_major_gemms"] + (1 if L_ >= 32 else 0)


@pytest.mark.parametrize("tokens,dim,inner", [(100, 64, 128), (300, 320, 1280), (77, 128, 192), (2048, 1280, 5120), (2000, 1280, 5120), (33000, 320, 1280)])
def test_feed_forward_geglu_fused(sd, oracle, gpu, rng, tokens, dim, inner):
    """FeedForward (block.hpp:193-247): Linear(dim, 2*inner) -> GEGLU -> Linear(inner, dim) + residual.  When 2*inner % 128 == 0 the
    first GEMM computes value and gate columns side by side and writes the f16 operand of the second one (no [tokens][2*inner]
    tensor); inner = 192 exercises the unfused fallback.  2048 / 2000 x 1280 -> 10240 (the SDXL 32x32-level FF1, full and with a ragged last row
    tile) takes the 256 x 320 tile with the weight image in the 16-column value / gate interleave (epi_geglu16: one v_permlane16_swap per
    register pair); 33000 x 320 the 256 x 128 tile with the 128-column pairing at a ragged row count."""
    x = rng.standard_normal((1, tokens, dim)).astype(np.float32)
    w1 = (rng.standard_normal((2 * inner, dim)) / np.sqrt(dim)).astype(np.float32)
    b1 = rng.standard_normal(2 * inner).astype(np.float32)
    w2 = (rng.standard_normal((dim, inner)) / np.sqrt(inner)).astype(np.float32)
    b2 = rng.standard_normal(dim).astype(np.float32)

    def build(g, L):
        xin = g.input(x)
        h = L.ggml_mul_mat(g.ctx, g.weight(w1, F16), xin)
        h = L.ggml_add_inplace(g.ctx, h, g.weight(b1, F32))
        ts = sd_tensor_nb(h)
        lo = L.ggml_view_4d(g.ctx, h, inner, tokens, 1, 1, ts[1], ts[2], ts[3], 0)
        hi = L.ggml_view_4d(g.ctx, h, inner, tokens, 1, 1, ts[1], ts[2], ts[3], inner * 4)
        gate = L.ggml_gelu_inplace(g.ctx, L.ggml_cont(g.ctx, hi))
        h = L.ggml_mul(g.ctx, lo, gate)
        y = L.ggml_mul_mat(g.ctx, g.weight(w2, F16), h)
        y = L.ggml_add_inplace(g.ctx, y, g.weight(b2, F32))
        return L.ggml_add(g.ctx, y, xin)

    before = sd.backend_stats() if _on_gpu() else None
    ref, out = run_both(sd, oracle, gpu, build)
    assert rel_l2(out, ref) < 1e-3   # oracle GELU goes through the f16 table
    if before is not None:
        fused = sd.backend_stats()["fused_linear_geglu"] - before["fused_linear_geglu"]
        assert fused == (1 if (2 * inner) % 128 == 0 else 0)


@pytest.mark.parametrize("dim", [0, 1, 2])
def test_concat_of_strided_head_views(sd, oracle, gpu, rng, dim):
    """concat over per-head views of a fused projection (flux.hpp:283-296, 540-544): inputs are strided [d, H, L, N] slices whose
    (d, H) plane is contiguous; dim 2 takes the plane-copy path, dims 0 / 1 the generic one."""
    d, H, N = 16, 3, 2
    La, Lb = 5, 9
    qa = rng.standard_normal((N, La, 3 * d * H)).astype(np.float32)
    qb = rng.standard_normal((N, Lb if dim == 2 else La, 3 * d * H)).astype(np.float32)

    def build(g, L):
        ta, tb = g.input(qa), g.input(qb)
        na, nb_ = sd_tensor_nb(ta), sd_tensor_nb(tb)
        va = L.ggml_view_4d(g.ctx, ta, d, H, qa.shape[1], N, 4 * d, na[1], na[2], 4 * d * H)       # the "k" third of a fused qkv
        vb = L.ggml_view_4d(g.ctx, tb, d, H, qb.shape[1], N, 4 * d, nb_[1], nb_[2], 4 * d * H)
        return L.ggml_concat(g.ctx, va, vb, dim)

    ref, out = run_both(sd, oracle, gpu, build)
    np.testing.assert_array_equal(out, ref)
    ka = qa[:, :, d * H:2 * d * H].reshape(N, qa.shape[1], H, d)
    kb = qb[:, :, d * H:2 * d * H].reshape(N, qb.shape[1], H, d)
    np.testing.assert_array_equal(out, np.concatenate([ka, kb], axis=3 - dim))


@pytest.mark.parametrize("ttype", [F32, F16, BF16, Q8_0, Q4_0])
def test_get_rows(sd, oracle, gpu, rng, ttype):
    """Embedding gather (CLIP / T5 token tables, T5 relative-attention bias): bit-exact against the oracle — both decode the same
    stored bytes — and against the numpy gather of the dequantised table for the non-quantised types."""
    table = rng.standard_normal((1000, 96)).astype(np.float32)
    ids = rng.integers(0, 1000, (2, 77)).astype(np.int32)
    ids[0, :3] = [0, 999, 0]

    def build(g, L):
        t = g.input(ids.reshape(-1), I32)
        t = L.ggml_reshape_3d(g.ctx, t, 154, 1, 1)              # ggml_extend.hpp:3577-3580: flattened ids [n_token * N, 1, 1]
        return L.ggml_get_rows(g.ctx, g.weight(table, ttype), t)

    ref, out = run_both(sd, oracle, gpu, build)
    np.testing.assert_array_equal(out, ref)
    if ttype in (F32, F16):
        np.testing.assert_array_equal(out.reshape(2, 77, 96), dequant(table, ttype)[ids])
    # a [heads, 32] bias table gathered by a [Lq*Lk] bucket list (t5.hpp:208-215)
    bias = rng.standard_normal((32, 4)).astype(np.float32)
    buckets = sd.t5_relative_position_buckets(19, 19).reshape(-1)
    ref, out = run_both(sd, oracle, gpu, lambda g, L: L.ggml_get_rows(g.ctx, g.weight(bias, F32), g.input(buckets, I32)))
    np.testing.assert_array_equal(out, ref)
    np.testing.assert_array_equal(out.reshape(-1, 4), bias[buckets])


def test_masked_attention_chain(sd, oracle, gpu, rng):
    """The text encoders' attention (ggml_extend.hpp:1460-1475 with a mask): scores + causal -inf mask (broadcast over heads) -> softmax
    -> V.  -inf entries must give exact zeros and no NaN."""
    H, Lq, d = 4, 77, 16
    q = rng.standard_normal((H, Lq, d)).astype(np.float32)
    k = rng.standard_normal((H, Lq, d)).astype(np.float32)
    v = rng.standard_normal((H, d, Lq)).astype(np.float32)
    mask = np.triu(np.full((Lq, Lq), -np.inf, dtype=np.float32), 1)

    def build(g, L):
        kq = L.ggml_mul_mat(g.ctx, g.input(k), g.input(q))
        kq = L.ggml_scale_inplace(g.ctx, kq, 0.25)
        kq = L.ggml_add_inplace(g.ctx, kq, g.input(mask))
        kq = L.ggml_soft_max_inplace(g.ctx, kq)
        return L.ggml_mul_mat(g.ctx, g.input(v), kq)

    ref, out = run_both(sd, oracle, gpu, build)
    assert np.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-4
