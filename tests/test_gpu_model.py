"""Whole-graph parity on the GPU: tiny-width UNet / VAE (same topology as SD1.5 / SDXL) through the C ABI,
MI355X backend vs the CPU oracle backend, same synthetic weights (seed 1234) and seeded inputs.

Tolerance: every contraction on both sides rounds activations to f16 and accumulates in f32, so single
layers agree to ~1e-4; through the ~60 layer deep UNet the summation-order noise compounds to ~1e-3 rel-L2.
Stated bar: rel-L2 <= 5e-3 per forward (f16), PSNR >= 35 dB on decoded pixels (SURVEY.md section 7 hard parts).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("model_name,flash", [("SD15_TINY", False), ("SD15_TINY", True), ("SDXL_TINY", False)])
def test_unet_forward_parity(sd, oracle, gpu, model_name, flash):
    model = getattr(sd, model_name)
    rng = np.random.default_rng(7)
    n = 2
    x = rng.standard_normal((n, 4, 32, 32)).astype(np.float32)
    t = np.array([731.0] * n, dtype=np.float32)
    ctx_dim = 64
    ctx = rng.standard_normal((1, 77, ctx_dim)).astype(np.float32)
    y = rng.standard_normal((1, 96)).astype(np.float32) if "XL" in model_name else None
    ref_e = sd.Engine(model=model, backend=oracle, flash_attn=flash)
    gpu_e = sd.Engine(model=model, backend=gpu, flash_attn=flash)
    ref = ref_e.unet_forward(x, t, ctx, y)
    out = gpu_e.unet_forward(x, t, ctx, y)
    assert np.isfinite(out).all()
    err = rel_l2(out, ref)
    print(f"{model_name} flash={flash}: rel-L2 {err:.3e}, nodes {gpu_e.stats()['graph_nodes']}")
    assert err < 5e-3
    # second call hits the plan cache and must give the same answer bit-for-bit
    out2 = gpu_e.unet_forward(x, t, ctx, y)
    np.testing.assert_array_equal(out, out2)


@pytest.mark.parametrize("n", [1, 3])
def test_unet_forward_parity_down_to_1x1_feature_maps(sd, oracle, gpu, n):
    """An 8x8 latent reaches 1x1 feature maps at the deepest UNet level, where every activation has the shape [1,1,C,N] of a bias / embedding
    operand: pattern matches keyed on shapes must not confuse them (round 2: the ResBlock skip ADD was claimed by two fused chains)."""
    rng = np.random.default_rng(70 + n)
    x = rng.standard_normal((n, 4, 8, 8)).astype(np.float32)
    t = np.full(n, 500.0, np.float32)
    ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
    ref = sd.Engine(model=sd.SD15_TINY, backend=oracle).unet_forward(x, t, ctx)
    out = sd.Engine(model=sd.SD15_TINY, backend=gpu).unet_forward(x, t, ctx)
    assert np.isfinite(out).all() and rel_l2(out, ref) < 5e-3


def test_vae_decode_parity(sd, oracle, gpu):
    rng = np.random.default_rng(8)
    z = rng.standard_normal((1, 4, 16, 16)).astype(np.float32) * 0.18215 * 3
    ref = sd.Engine(model=sd.SD15_TINY, backend=oracle).vae_decode(z)
    out = sd.Engine(model=sd.SD15_TINY, backend=gpu).vae_decode(z)
    mse = float(np.mean((out.astype(np.float64) - ref) ** 2))
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    print(f"VAE decode PSNR {psnr:.1f} dB")
    assert psnr > 35.0


def test_vae_decode_with_conv2d_scale_parity(sd, oracle, gpu):
    """SDXL engines run the VAE with Conv2d scale 1/32 like the reference without --vae (src/stable-diffusion.cpp:1477-1485): SCALE -> conv -> SCALE 1/s ->
    bias; the MI355X backend folds both SCALE nodes away (operand image x s, epilogue x 1/s).  Odd width: the 4-byte pack kernel; 2 images."""
    rng = np.random.default_rng(81)
    for shape in ((2, 4, 16, 16), (1, 4, 9, 7)):
        z = rng.standard_normal(shape).astype(np.float32) * 0.13025 * 3
        ref = sd.Engine(model=sd.SDXL_TINY, backend=oracle).vae_decode(z)
        e = sd.Engine(model=sd.SDXL_TINY, backend=gpu)
        st0 = sd.backend_stats() if gpu != oracle else None
        out = e.vae_decode(z)
        mse = float(np.mean((out.astype(np.float64) - ref) ** 2))
        psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
        print(f"VAE decode {shape} with Conv2d scale 1/32: PSNR {psnr:.1f} dB")
        assert np.isfinite(out).all() and psnr > 35.0
        if st0 is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
            st1 = sd.backend_stats()
            assert st1["fused_conv_scale"] - st0["fused_conv_scale"] == st1["fused_conv"] - st0["fused_conv"] > 20
        e.set_vae_conv2d_scale(1.0)   # --vae with a fixed VAE: plain convs
        plain = e.vae_decode(z)
        assert float(np.abs(plain - out).max()) < 2e-2
        np.testing.assert_array_equal(out.shape, plain.shape)


@pytest.mark.parametrize("model_name,zc,hw,n", [("SD15_TINY", 4, (12, 10), 3), ("SD15_TINY", 4, (9, 7), 2), ("SD15_TINY", 4, (64, 64), 2), ("SD35_TINY", 16, (128, 128), 1)])
def test_taesd_decode_parity(sd, oracle, gpu, model_name, zc, hw, n):
    """TAESD's decoder (src/model/vae/tae.hpp:123-183) through the C ABI on the GPU against the oracle backend: a ragged small latent, the
    512 x 512 and the 16-channel 1024 x 1024 decode (64-channel 3x3 convs + ReLU up to 1024 x 1024 feature maps).  The output is unclamped: compared as it leaves the graph."""
    rng = np.random.default_rng(23)
    z = (rng.standard_normal((n, zc) + hw) * 1.5).astype(np.float32)
    ref = sd.Engine(model=getattr(sd, model_name), backend=oracle).tae_decode(z)
    g = sd.Engine(model=getattr(sd, model_name), backend=gpu)
    out = g.tae_decode(z)
    err = rel_l2(out, ref)
    print(f"TAESD {model_name} {hw} x{n}: rel-L2 {err:.3e}, decode {g.stats()['last_decode_ms']:.2f} ms")
    assert np.isfinite(out).all() and err < 3e-3
    np.testing.assert_array_equal(g.tae_decode(z), out)   # plan cache / hipGraph replay


@pytest.mark.parametrize("model_name,hw,n", [("SD15_TINY", (128, 64), 2), ("SDXL_TINY", (128, 128), 1), ("SD35_TINY", (64, 96), 1)])
def test_vae_encode_and_img2img_parity(sd, oracle, gpu, model_name, hw, n):
    """sd_vae_encode (pad + stride-2 downsample convs, ragged sizes, the SDXL Conv2d scale, the 16-channel encoder) and an img2img trajectory started from its latent, GPU vs
    the oracle backend."""
    rng = np.random.default_rng(29)
    img = rng.random((n, 3) + hw).astype(np.float32)
    dit = model_name.startswith("SD35")
    cond = rng.standard_normal((1, 40, 96) if dit else (1, 77, 64)).astype(np.float32)
    y = rng.standard_normal((1, 64 if dit else 96)).astype(np.float32) if model_name != "SD15_TINY" else None
    res = []
    for be in (oracle, gpu):
        e = sd.Engine(model=getattr(sd, model_name), backend=be)
        lat, mom = e.vae_encode(img, seed=5, return_moments=True)
        traj = e.sample_latents(cond, cond * 0.5, width=hw[1], height=hw[0], steps=6, cfg=3.0, seed=8, batch=1, cond_y=y, uncond_y=y, init_latent=lat[0], strength=0.5,
                                fuse_cfg=True, device_sampler=True)
        res.append((lat, mom, traj))
    (lat_r, mom_r, tr_r), (lat_g, mom_g, tr_g) = res
    print(f"VAE encode {model_name} {hw} x{n}: moments rel-L2 {rel_l2(mom_g, mom_r):.2e}; img2img trajectory from its own latent {rel_l2(tr_g, tr_r):.2e}")
    assert np.isfinite(mom_g).all() and rel_l2(mom_g, mom_r) < 5e-3 and rel_l2(lat_g, lat_r) < 5e-3
    assert np.isfinite(tr_g).all() and rel_l2(tr_g, tr_r) < 2e-2


def test_activation_folded_into_the_conv_operand_image(sd, oracle, gpu):
    """conv -> ReLU -> conv (TAESD's blocks): a ReLU read only by convs is applied while their f16 NHWC operand image is written (option fuse_act_pack) — 20 unary
    launches fewer per decode, the in-place f32 result never written, bit-identical to the unfused plan."""
    if gpu == oracle:
        pytest.skip("planner behaviour of the MI355X backend")
    rng = np.random.default_rng(27)
    z = (rng.standard_normal((2, 4, 24, 20)) * 1.5).astype(np.float32)
    e = sd.Engine(model=sd.SD15_TINY, backend=gpu)

    def run():
        s0 = sd.backend_stats()
        out = e.tae_decode(z)
        s1 = sd.backend_stats()
        return out, s1["kernels_planned"] - s0["kernels_planned"]

    try:
        sd.backend_set_option("fuse_act_pack", 0)
        plain, k_plain = run()
    finally:
        sd.backend_set_option("fuse_act_pack", 1)
    fused, k_fused = run()
    print(f"TAESD decode: {k_plain} kernels unfused, {k_fused} with the activation folded into the operand image")
    np.testing.assert_array_equal(fused, plain)
    assert k_plain - k_fused == 28   # 20 ReLUs read only by convs lose their launch, 8 more (read by a conv and the residual ADD) merge with the pack pass


def test_skip_layer_guidance_parity(sd, oracle, gpu):
    """Skip-layer guidance on the GPU (SD3.5 tiny, CFG 3 + SLG on joint block 1 in a three-step window): the MMDiT forward without a block is its own graph / plan; trajectory and
    the skip forward alone against the oracle backend."""
    rng = np.random.default_rng(61)
    cond = rng.standard_normal((1, 40, 96)).astype(np.float32)
    uncond = rng.standard_normal((1, 40, 96)).astype(np.float32)
    y = rng.standard_normal((1, 64)).astype(np.float32)
    x = rng.standard_normal((2, 16, 14, 12)).astype(np.float32)
    t = np.array([731.0, 210.0], dtype=np.float32)
    kw = dict(width=64, height=96, steps=6, cfg=3.0, seed=8, batch=2, device_batch=2, cond_y=y, uncond_y=y, method=sd.EULER, fuse_cfg=True, slg=([1], 2.5, 0.2, 0.8))
    res = []
    for be in (oracle, gpu):
        e = sd.Engine(model=sd.SD35_TINY, backend=be, flash_attn=True)
        res.append((e.unet_forward_skip_layers(x, t, np.repeat(cond, 2, 0), np.repeat(y, 2, 0), [0, 2]), e.sample_latents(cond, uncond, **kw)))
    (f_r, tr_r), (f_g, tr_g) = res
    print(f"SLG: forward without blocks 0 and 2 rel-L2 {rel_l2(f_g, f_r):.2e}; CFG + SLG trajectory {rel_l2(tr_g, tr_r):.2e}")
    assert np.isfinite(f_g).all() and rel_l2(f_g, f_r) < 5e-3
    assert np.isfinite(tr_g).all() and rel_l2(tr_g, tr_r) < 2e-2


def test_sampler_trajectory_parity(sd, oracle, gpu):
    """4-step Euler-A with CFG 7, two images in one device batch vs the oracle's independent batch-1 runs."""
    rng = np.random.default_rng(9)
    cond = rng.standard_normal((1, 77, 64)).astype(np.float32)
    uncond = rng.standard_normal((1, 77, 64)).astype(np.float32)
    ref_e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    gpu_e = sd.Engine(model=sd.SD15_TINY, backend=gpu)
    kw = dict(width=128, height=128, steps=4, cfg=7.0, seed=42)
    out = gpu_e.sample_latents(cond, uncond, batch=2, device_batch=2, **kw)
    ref = np.concatenate([ref_e.sample_latents(cond, uncond, batch=1, seed=42 + b, width=128, height=128, steps=4, cfg=7.0) for b in range(2)])
    err = rel_l2(out, ref)
    print(f"trajectory rel-L2 {err:.3e}")
    assert err < 2e-2


@pytest.mark.parametrize("method,scheduler", [("DPMPP2M", "SCHED_KARRAS"), ("HEUN", "SCHED_KARRAS"), ("RES_2S", "SCHED_BETA"), ("EULER_A_CFG_PP", "SCHED_GITS"),
                                              ("LMS", "SCHED_BONG_TANGENT"), ("DPMPP2M_SDE_BT", "SCHEDULER_DEFAULT"), ("TCD", "SCHEDULER_DEFAULT")])
def test_more_samplers_trajectory_parity(sd, oracle, gpu, method, scheduler):
    """The multi-stage / multi-step samplers run the host loop (bit-exact against the reference's sample_k_diffusion on the CPU: tests/test_host_logic.py) around the DEVICE
    forward: 5-step trajectories with CFG 5, two images in one device batch, the cond / uncond pair in one graph, against the oracle backend's batch-1 runs — two-stage methods
    (second model call at an intermediate sigma), history methods, the CFG++ update that needs the unconditional prediction, the Brownian-tree noise."""
    rng = np.random.default_rng(19)
    cond = rng.standard_normal((1, 77, 64)).astype(np.float32)
    uncond = rng.standard_normal((1, 77, 64)).astype(np.float32)
    kw = dict(width=128, height=128, steps=5, cfg=5.0, method=getattr(sd, method), scheduler=getattr(sd, scheduler))
    gpu_e = sd.Engine(model=sd.SD15_TINY, backend=gpu)
    out = gpu_e.sample_latents(cond, uncond, batch=2, device_batch=2, seed=11, fuse_cfg=True, device_sampler=True, **kw)   # (device_sampler: falls back to the host loop)
    ref_e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    ref = np.concatenate([ref_e.sample_latents(cond, uncond, batch=1, seed=11 + b, **kw) for b in range(2)])
    err = rel_l2(out, ref)
    print(f"{method} / {scheduler}: trajectory rel-L2 {err:.3e}")
    assert np.isfinite(out).all() and err < 2e-2


@pytest.mark.parametrize("flash,wtype", [(False, "F16"), (True, "F16"), (True, "BF16")])
def test_mmdit_forward_parity(sd, oracle, gpu, flash, wtype):
    """SD3.5 MMDiT (tiny width, same topology incl. one MMDiT-X block; odd latent size exercises pad + crop) — SURVEY.md row a11.
    bf16 Linear weights (config 5) are decoded once into the f16 MFMA weight image; the oracle keeps them bf16 x f32-rounded-to-bf16
    like ggml-cpu, so that leg gets the looser bar."""
    rng = np.random.default_rng(17)
    n = 2
    x = rng.standard_normal((n, 16, 18, 15)).astype(np.float32)
    t = np.array([731.0, 210.0], dtype=np.float32)
    ctx = rng.standard_normal((1, 154, 96)).astype(np.float32)
    y = rng.standard_normal((1, 64)).astype(np.float32)
    wt = getattr(sd, wtype)
    ref = sd.Engine(model=sd.SD35_TINY, backend=oracle, flash_attn=flash, wtype=wt).unet_forward(x, t, ctx, y)
    gpu_e = sd.Engine(model=sd.SD35_TINY, backend=gpu, flash_attn=flash, wtype=wt)
    on_gpu = gpu != oracle
    before = sd.backend_stats() if on_gpu else None
    out = gpu_e.unet_forward(x, t, ctx, y)
    assert np.isfinite(out).all()
    err = rel_l2(out, ref)
    print(f"SD35_TINY flash={flash} {wtype}: rel-L2 {err:.3e}, nodes {gpu_e.stats()['graph_nodes']}")
    assert err < (2e-2 if wtype == "BF16" else 5e-3)
    if on_gpu and not os.environ.get("SDCPP_BACKEND_OPTS"):
        # the DiT fusions must actually be taken: LN+modulate -> operand image, gate+residual and GELU in the GEMM epilogue
        st = sd.backend_stats()
        d = {k: st[k] - before[k] for k in ("fused_modulate", "fused_gate", "fused_gelu", "fused_concat_heads")}
        print("DiT fusions taken:", d)
        assert d["fused_modulate"] >= 6   # LN+modulate in front of qkv / fc1 of both streams (the MMDiT-X block shares its LN: unfused)
        assert d["fused_gate"] >= 3       # attention projections (the tiny model's deep-K fc2 runs split-K and keeps the plain epilogue)
        assert d["fused_gelu"] >= 3
        assert d["fused_concat_heads"] >= 3   # joint-attention operands (a chain is left unfused when the address analysis cannot prove its inputs intact)
    np.testing.assert_array_equal(out, gpu_e.unet_forward(x, t, ctx, y))


def test_mmdit_without_qk_norm_forward_parity(sd, oracle, gpu):
    """SD3-medium's MMDiT variant (no qk-norm: the q / k parts of the fused qkv projection reach the joint attention without the per-head
    RMSNorm, mmdit.hpp:299-366 with qk_norm empty; no MMDiT-X block): the joint-attention operand pass runs with no norm weight on either
    stream — every projection goes through the arena-scratch redirect (plan_joint_qkv)."""
    rng = np.random.default_rng(19)
    x = rng.standard_normal((2, 16, 16, 12)).astype(np.float32)
    t = np.array([640.0, 333.0], dtype=np.float32)
    ctx = rng.standard_normal((1, 77, 96)).astype(np.float32)
    y = rng.standard_normal((1, 64)).astype(np.float32)
    ref = sd.Engine(model=sd.SD3M_TINY, backend=oracle, flash_attn=True).unet_forward(x, t, ctx, y)
    gpu_e = sd.Engine(model=sd.SD3M_TINY, backend=gpu, flash_attn=True)
    on_gpu = gpu != oracle
    before = sd.backend_stats() if on_gpu else None
    out = gpu_e.unet_forward(x, t, ctx, y)
    err = rel_l2(out, ref)
    print(f"SD3M_TINY (no qk-norm): rel-L2 {err:.3e}")
    assert np.isfinite(out).all() and err < 5e-3
    if on_gpu and not os.environ.get("SDCPP_BACKEND_OPTS"):
        st = sd.backend_stats()
        assert st["fused_joint_qkv"] - before["fused_joint_qkv"] == 6   # both streams of the three joint blocks
    np.testing.assert_array_equal(out, gpu_e.unet_forward(x, t, ctx, y))


def test_mmdit_flow_trajectory_and_vae_parity(sd, oracle, gpu):
    rng = np.random.default_rng(18)
    cond, uncond = (rng.standard_normal((1, 40, 96)).astype(np.float32) for _ in range(2))
    cy, uy = (rng.standard_normal((1, 64)).astype(np.float32) for _ in range(2))
    kw = dict(width=128, height=128, steps=4, cfg=4.5, method=sd.EULER, cond_y=cy, uncond_y=uy)
    gpu_e = sd.Engine(model=sd.SD35_TINY, backend=gpu, flash_attn=True)
    ref_e = sd.Engine(model=sd.SD35_TINY, backend=oracle, flash_attn=True)
    out = gpu_e.sample_latents(cond, uncond, batch=2, device_batch=2, seed=42, fuse_cfg=True, **kw)
    ref = np.concatenate([ref_e.sample_latents(cond, uncond, batch=1, seed=42 + b, **kw) for b in range(2)])
    err = rel_l2(out, ref)
    print(f"SD3.5 flow trajectory rel-L2 {err:.3e}")
    assert err < 2e-2
    z = ref[:1]
    a, b = gpu_e.vae_decode(z), ref_e.vae_decode(z)
    psnr = 10 * np.log10(1.0 / max(float(np.mean((a.astype(np.float64) - b) ** 2)), 1e-20))
    print(f"16-channel VAE decode PSNR {psnr:.1f} dB")
    assert psnr > 35.0


@pytest.mark.parametrize("flash,wtype", [(True, "F16"), (False, "F16"), (True, "Q4_0")])
def test_flux_forward_parity(sd, oracle, gpu, flash, wtype):
    """FLUX (tiny width, same topology: double + single stream blocks, RoPE node chain, fused qkv+mlp linear1) — SURVEY.md row a12.
    q4_0 Linear weights (config 4): the GPU multiplies f16-rounded activations with the exactly dequantised weights, the oracle
    (like ggml-cpu) quantises activations to q8_0 first — the looser bar covers that."""
    rng = np.random.default_rng(27)
    x = rng.standard_normal((2, 16, 18, 15)).astype(np.float32)
    t = np.array([0.81, 0.27], dtype=np.float32)
    ctx = rng.standard_normal((1, 40, 96)).astype(np.float32)
    y = rng.standard_normal((1, 64)).astype(np.float32)
    wt = getattr(sd, wtype)
    ref = sd.Engine(model=sd.FLUX_TINY, backend=oracle, flash_attn=flash, wtype=wt).unet_forward(x, t, ctx, y)
    gpu_e = sd.Engine(model=sd.FLUX_TINY, backend=gpu, flash_attn=flash, wtype=wt)
    before = sd.backend_stats() if gpu != oracle else None
    out = gpu_e.unet_forward(x, t, ctx, y)
    assert np.isfinite(out).all()
    if before is not None and not os.environ.get("SDCPP_BACKEND_OPTS"):
        # every apply_rope node chain (q and k of 2 double + 2 single blocks) must have been replaced by the rotary kernel
        assert sd.backend_stats()["fused_rope"] - before["fused_rope"] == 8
    err = rel_l2(out, ref)
    print(f"FLUX_TINY flash={flash} {wtype}: rel-L2 {err:.3e}, nodes {gpu_e.stats()['graph_nodes']}")
    assert err < (6e-2 if wtype == "Q4_0" else 5e-3)
    np.testing.assert_array_equal(out, gpu_e.unet_forward(x, t, ctx, y))


def test_flux_trajectory_parity(sd, oracle, gpu):
    rng = np.random.default_rng(28)
    cond = rng.standard_normal((1, 24, 96)).astype(np.float32)
    cy = rng.standard_normal((1, 64)).astype(np.float32)
    kw = dict(width=128, height=128, steps=4, cfg=1.0, method=sd.EULER, cond_y=cy)
    out = sd.Engine(model=sd.FLUX_TINY, backend=gpu, flash_attn=True).sample_latents(cond, None, batch=2, device_batch=2, seed=42, **kw)
    ref_e = sd.Engine(model=sd.FLUX_TINY, backend=oracle, flash_attn=True)
    ref = np.concatenate([ref_e.sample_latents(cond, None, batch=1, seed=42 + b, **kw) for b in range(2)])
    err = rel_l2(out, ref)
    print(f"FLUX flow trajectory rel-L2 {err:.3e}")
    assert err < 2e-2


# ---- text encoders -------------------------------------------------------------------------------------
def _prompt(rng, n_words, pad, vocab=1000, length=77):
    ids = np.full(length, pad, dtype=np.int32)
    ids[0] = vocab - 2
    ids[1:1 + n_words] = rng.integers(1, vocab - 2, n_words)
    ids[1 + n_words] = vocab - 1
    return ids


@pytest.mark.parametrize("wtype", ["F16", "Q8_0"])
def test_clip_text_towers_parity(sd, oracle, gpu, wtype):
    """CLIP ViT-L and bigG text towers (tiny width, same topology: GET_ROWS embeddings, causal-mask attention, quick-GELU / GELU MLP,
    penultimate-layer output, pooled + text_projection) on the GPU vs the oracle.  Same f16 bar as the UNet: rel-L2 <= 5e-3; q8_0
    weights get the looser bar because the oracle also quantises activations."""
    rng = np.random.default_rng(31)
    wt = getattr(sd, wtype)
    tol = 5e-3 if wtype == "F16" else 3e-2
    ref_e = sd.Engine(model=sd.SDXL_TINY, backend=oracle, wtype=wt)
    gpu_e = sd.Engine(model=sd.SDXL_TINY, backend=gpu, wtype=wt)
    ids_l, ids_g = _prompt(rng, 9, 999), _prompt(rng, 9, 0)
    for which, ids in ((0, ids_l), (1, ids_g)):
        a, b = gpu_e.clip_forward(which, ids, clip_skip=2), ref_e.clip_forward(which, ids, clip_skip=2)
        assert np.isfinite(a).all()
        print(f"clip tower {which} {wtype}: rel-L2 {rel_l2(a, b):.3e}")
        assert rel_l2(a, b) < tol
    pa = gpu_e.clip_forward(1, ids_g, max_token_idx=10, return_pooled=True)
    pb = ref_e.clip_forward(1, ids_g, max_token_idx=10, return_pooled=True)
    assert rel_l2(pa, pb) < tol
    # SD1.x flavour: all layers + final LN
    a = sd.Engine(model=sd.SD15_TINY, backend=gpu, wtype=wt).clip_forward(0, ids_l)
    b = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=wt).clip_forward(0, ids_l)
    assert rel_l2(a, b) < tol


def test_t5_encoder_and_conditioner_parity(sd, oracle, gpu):
    """T5 encoder stack (relative-position bias gather, RMS norm, gated-GELU FF with the 1/32 pre-scale) and the SD3 conditioner
    composition (two CLIP towers + T5 -> [154, ctx] context and the pooled vector) on the GPU vs the oracle."""
    rng = np.random.default_rng(32)
    ref_e = sd.Engine(model=sd.SD35_TINY, backend=oracle)
    gpu_e = sd.Engine(model=sd.SD35_TINY, backend=gpu)
    for n in (77, 24):
        ids = rng.integers(0, 1000, n).astype(np.int32)
        a, b = gpu_e.t5_forward(ids), ref_e.t5_forward(ids)
        assert np.isfinite(a).all()
        print(f"T5 n={n}: rel-L2 {rel_l2(a, b):.3e}")
        assert rel_l2(a, b) < 5e-3
    il, ig, it = _prompt(rng, 6, 999), _prompt(rng, 6, 0), rng.integers(0, 1000, 77).astype(np.int32)
    w = np.ones(77, np.float32)
    w[2:6] = 1.4
    (ca, ya), (cb, yb) = gpu_e.get_learned_condition((il, w), ig, it), ref_e.get_learned_condition((il, w), ig, it)
    assert ca.shape == (1, 154, 96) and ya.shape == (1, 64)
    assert rel_l2(ca, cb) < 5e-3 and rel_l2(ya, yb) < 5e-3


def test_tokens_to_image_on_gpu(sd, oracle, gpu):
    """ids -> conditioner -> 3 Euler-A steps -> VAE decode, all on the GPU engine; pixels vs the same pipeline on the oracle."""
    rng = np.random.default_rng(33)
    ids = _prompt(rng, 8, 999)
    empty = _prompt(rng, 0, 999)
    imgs = []
    for dev in (gpu, oracle):
        e = sd.Engine(model=sd.SD15_TINY, backend=dev)
        c, _ = e.get_learned_condition(ids)
        u, _ = e.get_learned_condition(empty)
        imgs.append(e.generate_image(c, u, width=64, height=64, steps=3, cfg=5.0, seed=11).astype(np.float64) / 255.0)
    mse = float(np.mean((imgs[0] - imgs[1]) ** 2))
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    print(f"tokens -> image PSNR {psnr:.1f} dB")
    assert psnr > 30.0


@pytest.mark.parametrize("model_name,method,cfg", [("SD15_TINY", "EULER_A", 7.0), ("SD35_TINY", "EULER", 4.5)])
def test_device_resident_sampler_parity(sd, oracle, gpu, model_name, method, cfg):
    """the whole iteration (x*c_in, cond+uncond model pair, CFG combine, Euler(-A) update, noise add) as one
    graph per step on latents that stay in a backend buffer, queued without host synchronisation — against the host-side loop on the
    same device (same kernels for the network, f32 elementwise math in the same order: rel-L2 <= 1e-5) and against the oracle."""
    rng = np.random.default_rng(41)
    dit = model_name.startswith("SD35")
    cond = rng.standard_normal((1, 40 if dit else 77, 96 if dit else 64)).astype(np.float32)
    uncond = rng.standard_normal(cond.shape).astype(np.float32)
    cy, uy = ((rng.standard_normal((1, 64)).astype(np.float32) for _ in range(2)) if dit else (None, None))
    kw = dict(width=128, height=128, steps=4, cfg=cfg, seed=5, batch=2, device_batch=2, fuse_cfg=True, method=getattr(sd, method), cond_y=cy, uncond_y=uy)
    e = sd.Engine(model=getattr(sd, model_name), backend=gpu, flash_attn=True)
    host = e.sample_latents(cond, uncond, **kw)
    dev = e.sample_latents(cond, uncond, device_sampler=True, **kw)
    assert np.isfinite(dev).all()
    print(f"{model_name} device-resident vs host loop: rel-L2 {rel_l2(dev, host):.3e}")
    assert rel_l2(dev, host) < 1e-5
    np.testing.assert_array_equal(dev, e.sample_latents(cond, uncond, device_sampler=True, **kw))   # state buffer and plan reuse
    ref = sd.Engine(model=getattr(sd, model_name), backend=oracle, flash_attn=True).sample_latents(cond, uncond, **kw)
    assert rel_l2(dev, ref) < 2e-2
