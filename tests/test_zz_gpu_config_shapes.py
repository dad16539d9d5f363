"""Parity at the BENCHMARKED shapes of BASELINE.json configs 3, 4, 5 and of the 1024x1024 VAE decode ("next round" item 1).

bench.py quotes throughput on SDXL at latent 128x128, FLUX.1-dev at 4096 + 256 tokens, SD3.5-large at 4096 + 154 tokens and the KL-VAE at
128x128 -> 1024x1024; every contraction shape those graphs launch is checked here against float64 products recomputed from the SAME
rounded operands the MFMA path consumes (f16-rounded activations; weights exactly as stored: f16, bf16, dequantised q8_0 / q4_0 rounded to
f16), at randomly sampled output elements — no oracle run needed, so the full sizes are affordable:

  reference shapes: src/model/diffusion/unet.hpp:47-57 (SDXL), flux.hpp:430-700 (FLUX blocks: 3072 -> 9216 / 12288 / 21504, 15360 -> 3072,
  24 heads x 128), mmdit.hpp:614-699 (SD3.5: 2432 -> 7296 / 9728, 38 heads x 64), auto_encoder_kl.hpp:444-492 (VAE decoder at 128^2 .. 1024^2,
  mid attention 1 head x 512 over 16384 positions).

Bars: 1e-3 of the output scale for f32-accumulated contractions (K <= 15360), 3e-3 for attention rows (P and Q enter the MFMA as f16).
test_zz_gpu_fullsize.py carries the SD1.5 (configs 1 / 2) shapes and the oracle-backed whole-graph checks.
"""
import os

import numpy as np
import pytest

from ggml_graph import BF16, F16, F32, Q4_0, Q8_0, Graph, dequant, to_bf16_bits

pytestmark = pytest.mark.gpu

ON_GPU = os.environ.get("SDCPP_GPU_TESTS_ON_ORACLE") != "1"


def f16r(a):
    return np.asarray(a).astype(np.float16).astype(np.float64)


def run(dev, build):
    with Graph(dev) as g:
        return g.run(build(g, g.L))


def stored_weight(w, wtype):
    """The values the stored weight decodes to, as the MFMA weight image holds them (rounded to f16), float64."""
    if wtype == BF16:
        u = to_bf16_bits(w).astype(np.uint32) << 16
        return f16r(u.view(np.float32).reshape(w.shape))
    return f16r(dequant(w, wtype))


# (tag, tokens, K, M, weight type)
LINEAR_CASES = [
    # config 3: SDXL, cond+uncond pair of one image: 2 x 4096 tokens at 640, 2 x 1024 at 1280 (q8_0 Linear)
    ("sdxl", 8192, 640, 640, Q8_0), ("sdxl", 8192, 640, 5120, Q8_0), ("sdxl", 8192, 2560, 640, Q8_0),
    ("sdxl", 2048, 1280, 1280, Q8_0), ("sdxl", 2048, 1280, 10240, Q8_0), ("sdxl", 2048, 5120, 1280, Q8_0), ("sdxl", 154, 2048, 1280, Q8_0),
    # config 4: FLUX.1-dev, 4096 image + 256 text tokens, q4_0 (above 512 rows: f16 weight image) and the same shapes in f16
    ("flux", 4352, 3072, 21504, F16), ("flux", 4352, 3072, 21504, Q4_0), ("flux", 4096, 3072, 9216, Q4_0), ("flux", 4096, 3072, 12288, Q4_0),
    ("flux", 4096, 12288, 3072, Q4_0), ("flux", 4352, 15360, 3072, Q4_0), ("flux", 256, 3072, 9216, Q4_0),
    # config 5: SD3.5-large, 4096 image + 154 context tokens, bf16
    ("sd35", 4096, 2432, 7296, BF16), ("sd35", 4096, 2432, 9728, BF16), ("sd35", 4096, 9728, 2432, BF16), ("sd35", 154, 2432, 7296, BF16),
]
if not ON_GPU:
    LINEAR_CASES = [("sdxl", 300, 640, 640, Q8_0), ("flux", 200, 512, 768, Q4_0), ("sd35", 154, 256, 320, BF16)]


@pytest.mark.parametrize("tag,tokens,K,M,wtype", LINEAR_CASES)
def test_config_linear(sd, gpu, tag, tokens, K, M, wtype):
    rng = np.random.default_rng(1000 + K + M + tokens)
    x = rng.standard_normal((tokens, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)

    def build(g, L):
        return L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, g.weight(w, wtype), g.input(x)), g.weight(b, F32))

    out = run(gpu, build).reshape(tokens, M)
    assert np.isfinite(out).all()
    ws = stored_weight(w, wtype)
    rows = np.unique(np.concatenate([rng.integers(0, tokens, 40), [0, tokens - 1]]))   # incl. the last (possibly ragged) row tile
    xr = f16r(x[rows]) if ON_GPU else x[rows].astype(np.float64)
    ref = xr @ ws.T + b
    tol = 1e-3 if ON_GPU else (3e-2 if wtype in (Q8_0, Q4_0) else 1e-2)   # self-check mode: the oracle quantises activations like ggml-cpu
    assert np.abs(out[rows] - ref).max() < tol * max(1.0, float(np.abs(ref).max())), tag
    np.testing.assert_array_equal(out, run(gpu, build).reshape(tokens, M))


# (tag, N, IC, OC, HW, ks, stride, upscale)
CONV_CASES = [
    ("sdxl", 2, 320, 320, 128, 3, 1, False), ("sdxl", 2, 640, 640, 64, 3, 1, False), ("sdxl", 2, 1280, 1280, 32, 3, 1, False),
    ("sdxl", 2, 320, 320, 128, 3, 2, False), ("sdxl", 2, 960, 320, 128, 3, 1, False), ("sdxl", 2, 1920, 640, 64, 3, 1, False),
    ("sdxl", 2, 640, 640, 32, 3, 1, True),
    # KL-VAE decoder at 128x128 -> 1024x1024 (one image): 512 ch @128^2 / 256^2, 256 ch @512^2, 128 ch @1024^2 is 537 MB per map: sampled at 512^2 + 1024^2
    ("vae", 1, 512, 512, 128, 3, 1, False), ("vae", 1, 512, 512, 128, 3, 1, True), ("vae", 1, 512, 256, 512, 3, 1, False),
    ("vae", 1, 256, 256, 512, 3, 1, False), ("vae", 1, 256, 256, 512, 3, 1, True), ("vae", 1, 128, 128, 1024, 3, 1, False), ("vae", 1, 128, 3, 1024, 3, 1, False),
    ("vae", 1, 16, 512, 128, 3, 1, False),
]
if not ON_GPU:
    CONV_CASES = [("sdxl", 1, 64, 64, 16, 3, 1, True), ("vae", 1, 16, 64, 16, 3, 1, False)]


@pytest.mark.parametrize("tag,N,IC,OC,HW,ks,stride,ups", CONV_CASES)
def test_config_conv(sd, gpu, tag, N, IC, OC, HW, ks, stride, ups):
    rng = np.random.default_rng(2000 + IC + OC + HW + int(ups))
    x = rng.standard_normal((N, IC, HW, HW), dtype=np.float32)
    w = (rng.standard_normal((OC, IC, ks, ks)) / np.sqrt(IC * ks * ks)).astype(np.float32)
    b = rng.standard_normal(OC).astype(np.float32)
    pad = ks // 2

    def build(g, L):
        xin = g.input(x)
        if ups:
            xin = L.ggml_upscale(g.ctx, xin, 2, 0)   # nearest x2 in front of the conv (UpSampleBlock, block.hpp:57-64): an index shift in the gather
        y = L.ggml_conv_2d(g.ctx, g.weight(w, F16), xin, stride, stride, pad, pad, 1, 1)
        return L.ggml_add_inplace(g.ctx, y, L.ggml_reshape_4d(g.ctx, g.weight(b, F32), 1, 1, OC, 1))

    out = run(gpu, build)
    S = HW * 2 if ups else HW
    O = (S + 2 * pad - ks) // stride + 1
    assert out.shape == (N, OC, O, O) and np.isfinite(out).all()
    w16 = f16r(w)
    scale = float(np.abs(out[:, :, ::max(1, O // 64), ::max(1, O // 64)]).mean())

    def ref_at(n, oc, oh, ow):
        acc = float(b[oc])
        for kh in range(ks):
            for kw in range(ks):
                ih, iw = oh * stride + kh - pad, ow * stride + kw - pad
                if 0 <= ih < S and 0 <= iw < S:
                    sy, sx = (ih // 2, iw // 2) if ups else (ih, iw)
                    acc += float(np.dot(f16r(x[n, :, sy, sx]), w16[oc, :, kh, kw]))
        return acc

    pts = [(rng.integers(N), rng.integers(OC), rng.integers(O), rng.integers(O)) for _ in range(64)]
    pts += [(0, min(1, OC - 1), oh, ow) for (oh, ow) in ((0, 0), (0, O - 1), (O - 1, 0), (O - 1, O - 1))]   # zero-page taps
    for (n, oc, oh, ow) in pts:
        ref = ref_at(n, oc, oh, ow)
        assert abs(out[n, oc, oh, ow] - ref) < 1e-3 * max(scale, abs(ref)), (tag, n, oc, oh, ow, out[n, oc, oh, ow], ref)
    if OC * O * O * N <= (1 << 27):
        np.testing.assert_array_equal(out, run(gpu, build))


# (tag, d, Lq, Lk, heads x images)
FLASH_CASES = [
    ("sdxl", 64, 4096, 4096, 20), ("sdxl", 64, 1024, 1024, 40), ("sdxl", 64, 4096, 77, 20), ("sdxl", 64, 1024, 77, 40),
    ("flux", 128, 4352, 4352, 24),
    ("sd35", 64, 4250, 4250, 38),
]
if not ON_GPU:
    FLASH_CASES = [("sdxl", 64, 256, 77, 2), ("flux", 128, 200, 200, 2)]


@pytest.mark.parametrize("tag,d,Lq,Lk,HN", FLASH_CASES)
def test_config_flash_attention(sd, gpu, tag, d, Lq, Lk, HN):
    rng = np.random.default_rng(3000 + d + Lk + HN)
    q = rng.standard_normal((HN, Lq, d)).astype(np.float32)
    k = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    v = rng.standard_normal((HN, Lk, d)).astype(np.float32)
    sc = 1.0 / np.sqrt(d)

    def build(g, L):
        return L.ggml_flash_attn_ext(g.ctx, g.input(q), g.input(k, F16), g.input(v, F16), None, sc, 0.0, 0.0)

    out = run(gpu, build)            # [1, Lq, HN, d]
    assert out.shape == (1, Lq, HN, d) and np.isfinite(out).all()
    k16, v16 = f16r(k), f16r(v)
    picks = [(rng.integers(HN), rng.integers(Lq)) for _ in range(32)] + [(HN - 1, Lq - 1), (0, 0)]   # incl. the ragged last query block
    for h, i in picks:
        s = (k16[h] @ q[h, i].astype(np.float64)) * sc
        p = np.exp(s - s.max())
        ref = (p / p.sum()) @ v16[h]
        tol = 3e-3 if ON_GPU else 3e-2
        assert np.abs(out[0, i, h] - ref).max() < tol * max(1.0, float(np.abs(ref).max())), (tag, h, i)
    np.testing.assert_array_equal(out, run(gpu, build))


@pytest.mark.parametrize("L_", [16384] if ON_GPU else [256])
def test_config_vae_mid_attention_d512(sd, gpu, L_):
    """KL-VAE mid-block attention at 128x128 latents (auto_encoder_kl.hpp:104-159): ONE head of d = 512 over 16384 positions, as the
    MUL_MAT -> SCALE -> SOFT_MAX -> MUL_MAT chain the reference emits without the flash flag; the backend composes it from MFMA GEMMs over
    f16 images with an f16 row softmax between them (536 MB of scores)."""
    d = 512
    rng = np.random.default_rng(3500)
    q = rng.standard_normal((1, L_, d)).astype(np.float32)
    k = rng.standard_normal((1, L_, d)).astype(np.float32)
    vt = rng.standard_normal((1, d, L_)).astype(np.float32)
    sc = 1.0 / np.sqrt(d)

    def build(g, L):
        kq = L.ggml_mul_mat(g.ctx, g.input(k), g.input(q))
        kq = L.ggml_scale_inplace(g.ctx, kq, sc)
        kq = L.ggml_soft_max_inplace(g.ctx, kq)
        return L.ggml_mul_mat(g.ctx, g.input(vt), kq)

    before = sd.backend_stats() if ON_GPU else None
    out = run(gpu, build)            # [1, 1, L, d]
    assert out.shape == (1, 1, L_, d) and np.isfinite(out).all()
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        assert sd.backend_stats()["gemm_attention"] - before["gemm_attention"] == 1
    k16, v16 = f16r(k[0]), f16r(vt[0])
    for i in list(rng.integers(0, L_, 24)) + [0, L_ - 1]:
        s = (k16 @ f16r(q[0, i])) * sc
        p = np.exp(s - s.max())
        ref = v16 @ (p / p.sum())
        assert np.abs(out[0, 0, i] - ref).max() < 4e-3 * max(1.0, float(np.abs(ref).max())), i
