"""oracle/torch_ref.py — TEST INFRASTRUCTURE ONLY.  *** PARITY UNPINNED (see oracle/ggml_cpu_ref.cpp header) ***

Independent PyTorch-CPU fp32 restatement of the model MATH on the hot path (no ggml, no graph IR): the SD1.x /
SDXL UNet eps-prediction and the KL-VAE decoder, driven by the engine's named weights.  It cross-checks the
graph BUILDERS (csrc/host/nn.hpp, models.hpp) — the C++ oracle shares those with the product, so a wrong
topology would otherwise go unnoticed — and is the "mathematical definition" leg of the three-way comparison
in SURVEY.md section 8(c): C++ oracle (ggml-cpu rounding) <-> PyTorch fp32 <-> HIP kernels.

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class Weights:
    """name -> torch tensor, fetched lazily from an sdcpp_amd.Engine (dequantised to f32)."""

    def __init__(self, engine, prefix: str):
        self.e = engine
        self.prefix = prefix
        self.cache = {}

    def __call__(self, name: str) -> torch.Tensor:
        full = self.prefix + name
        if full not in self.cache:
            self.cache[full] = torch.from_numpy(self.e.get_tensor(full).copy())
        return self.cache[full]

    def has(self, name: str) -> bool:
        try:
            self.e.tensor_info(self.prefix + name)
            return True
        except KeyError:
            return False

    def sub(self, more: str) -> "Weights":
        w = Weights(self.e, self.prefix + more)
        w.cache = self.cache
        return w


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    # src/core/ggml_extend.hpp:1579-1606 (cos first, then sin)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def linear(w: Weights, x):
    return F.linear(x, w("weight"), w("bias") if w.has("bias") else None)


def conv(w: Weights, x, stride=1, padding=0):
    return F.conv2d(x, w("weight"), w("bias"), stride=stride, padding=padding)


def group_norm(w: Weights, x):
    return F.group_norm(x, 32, w("weight"), w("bias"), eps=1e-6)  # ggml_extend.hpp:1502-1520


def layer_norm(w: Weights, x):
    return F.layer_norm(x, (x.shape[-1],), w("weight"), w("bias"), eps=1e-5)  # ggml_extend.hpp:1487-1500


def res_block(w: Weights, x, emb):
    # src/model/common/block.hpp:126-179
    h = conv(w.sub("in_layers.2."), F.silu(group_norm(w.sub("in_layers.0."), x)), padding=1)
    e = linear(w.sub("emb_layers.1."), F.silu(emb))
    h = h + e[:, :, None, None]
    h = conv(w.sub("out_layers.3."), F.silu(group_norm(w.sub("out_layers.0."), h)), padding=1)
    if w.has("skip_connection.weight"):
        x = conv(w.sub("skip_connection."), x)
    return h + x


def attention(q, k, v, n_head):
    # ggml_ext_attention_ext, ggml_extend.hpp:1349-1485: scale 1/sqrt(d_head), no mask
    B, Lq, C = q.shape
    d = C // n_head
    q = q.view(B, Lq, n_head, d).transpose(1, 2)
    k = k.view(B, -1, n_head, d).transpose(1, 2)
    v = v.view(B, -1, n_head, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    return o.transpose(1, 2).reshape(B, Lq, C)


def cross_attention(w: Weights, x, context, n_head):
    # block.hpp:354-393
    q = linear(w.sub("to_q."), x)
    k = linear(w.sub("to_k."), context)
    v = linear(w.sub("to_v."), context)
    return linear(w.sub("to_out.0."), attention(q, k, v, n_head))


def feed_forward(w: Weights, x):
    # GEGLU: block.hpp:193-210 (x * gelu(gate), tanh approximation), FeedForward :291-304
    h = linear(w.sub("net.0.proj."), x)
    a, gate = h.chunk(2, dim=-1)
    return linear(w.sub("net.2."), a * F.gelu(gate, approximate="tanh"))


def basic_transformer_block(w: Weights, x, context, n_head):
    # block.hpp:427-466
    x = cross_attention(w.sub("attn1."), layer_norm(w.sub("norm1."), x), layer_norm(w.sub("norm1."), x), n_head) + x
    x = cross_attention(w.sub("attn2."), layer_norm(w.sub("norm2."), x), context, n_head) + x
    return feed_forward(w.sub("ff."), layer_norm(w.sub("norm3."), x)) + x


def spatial_transformer(w: Weights, x, context, n_head, depth, use_linear):
    # block.hpp:528-577
    x_in = x
    B, C, H, W = x.shape
    x = group_norm(w.sub("norm."), x)
    if use_linear:
        x = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
        x = linear(w.sub("proj_in."), x)
    else:
        x = conv(w.sub("proj_in."), x)
        x = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
    for i in range(depth):
        x = basic_transformer_block(w.sub(f"transformer_blocks.{i}."), x, context, n_head)
    if use_linear:
        x = linear(w.sub("proj_out."), x)
        x = x.reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        x = x.reshape(B, H, W, C).permute(0, 3, 1, 2)
        x = conv(w.sub("proj_out."), x)
    return x + x_in


UNET_CFG = {
    # name: (model_channels, channel_mult, attention_resolutions, transformer_depth, num_heads, num_head_channels, use_linear, sdxl)
    "SD15": (320, [1, 2, 4, 4], [4, 2, 1], [1, 1, 1, 1], 8, -1, False, False),
    "SDXL": (320, [1, 2, 4], [4, 2], [1, 2, 10], -1, 64, True, True),
    "SD15_TINY": (32, [1, 2, 4, 4], [4, 2, 1], [1, 1, 1, 1], 2, -1, False, False),
    "SDXL_TINY": (32, [1, 2, 4], [4, 2], [1, 1, 2], -1, 16, True, True),
}


def unet_forward(engine, model: str, x, timesteps, context, y=None):
    """UnetModelBlock::forward — src/model/diffusion/unet.hpp:526-745.  x [N,C,H,W], context [N|1,77,D], y [N|1,adm]."""
    mc, mult, attn_res, depth, num_heads, nhc, use_linear, sdxl = UNET_CFG[model]
    w = Weights(engine, "model.diffusion_model.")
    x = torch.as_tensor(x, dtype=torch.float32)
    context = torch.as_tensor(context, dtype=torch.float32)
    N = x.shape[0]
    if context.shape[0] != N:
        context = context.expand(N, -1, -1)
    emb = timestep_embedding(torch.as_tensor(timesteps, dtype=torch.float32), mc)
    emb = linear(w.sub("time_embed.2."), F.silu(linear(w.sub("time_embed.0."), emb)))
    if sdxl:
        yy = torch.as_tensor(y, dtype=torch.float32)
        if yy.shape[0] != N:
            yy = yy.expand(N, -1)
        emb = emb + linear(w.sub("label_emb.0.2."), F.silu(linear(w.sub("label_emb.0.0."), yy)))

    def heads(ch):
        return (num_heads if nhc == -1 else ch // nhc)

    hs = []
    h = conv(w.sub("input_blocks.0.0."), x, padding=1)
    hs.append(h)
    idx, ds, ch = 0, 1, mc
    for i, m in enumerate(mult):
        for _ in range(2):
            idx += 1
            h = res_block(w.sub(f"input_blocks.{idx}.0."), h, emb)
            ch = m * mc
            if ds in attn_res:
                h = spatial_transformer(w.sub(f"input_blocks.{idx}.1."), h, context, heads(ch), depth[i], use_linear)
            hs.append(h)
        if i != len(mult) - 1:
            idx += 1
            h = conv(w.sub(f"input_blocks.{idx}.0.op."), h, stride=2, padding=1)
            hs.append(h)
            ds *= 2
    h = res_block(w.sub("middle_block.0."), h, emb)
    h = spatial_transformer(w.sub("middle_block.1."), h, context, heads(ch), depth[-1], use_linear)
    h = res_block(w.sub("middle_block.2."), h, emb)
    oidx = 0
    for i in reversed(range(len(mult))):
        for j in range(3):
            h = torch.cat([h, hs.pop()], dim=1)
            h = res_block(w.sub(f"output_blocks.{oidx}.0."), h, emb)
            ch = mult[i] * mc
            up_idx = 1
            if ds in attn_res:
                h = spatial_transformer(w.sub(f"output_blocks.{oidx}.1."), h, context, heads(ch), depth[i], use_linear)
                up_idx += 1
            if i > 0 and j == 2:
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = conv(w.sub(f"output_blocks.{oidx}.{up_idx}.conv."), h, padding=1)
                ds //= 2
            oidx += 1
    h = F.silu(group_norm(w.sub("out.0."), h))
    return conv(w.sub("out.2."), h, padding=1).numpy()


def vae_resnet(w: Weights, x):
    # src/model/vae/auto_encoder_kl.hpp:10-60
    h = conv(w.sub("conv1."), F.silu(group_norm(w.sub("norm1."), x)), padding=1)
    h = conv(w.sub("conv2."), F.silu(group_norm(w.sub("norm2."), h)), padding=1)
    if w.has("nin_shortcut.weight"):
        x = conv(w.sub("nin_shortcut."), x)
    return h + x


def vae_attn(w: Weights, x):
    # auto_encoder_kl.hpp:62-159 (conv projections, single head)
    B, C, H, W = x.shape
    h = group_norm(w.sub("norm."), x)
    q = conv(w.sub("q."), h).permute(0, 2, 3, 1).reshape(B, H * W, C)
    k = conv(w.sub("k."), h).permute(0, 2, 3, 1).reshape(B, H * W, C)
    v = conv(w.sub("v."), h).permute(0, 2, 3, 1).reshape(B, H * W, C)
    o = attention(q, k, v, 1).reshape(B, H, W, C).permute(0, 3, 1, 2)
    return conv(w.sub("proj_out."), o) + x


def vae_decode(engine, latents, scale_factor=0.18215, ch_mult=(1, 2, 4, 4)):
    """decode_first_stage + Decoder::forward — stable-diffusion.cpp:3062-3078; auto_encoder_kl.hpp:444-492, 589-620."""
    w = Weights(engine, "first_stage_model.")
    z = torch.as_tensor(latents, dtype=torch.float32) / scale_factor
    z = conv(w.sub("post_quant_conv."), z)
    d = w.sub("decoder.")
    h = conv(d.sub("conv_in."), z, padding=1)
    h = vae_resnet(d.sub("mid.block_1."), h)
    h = vae_attn(d.sub("mid.attn_1."), h)
    h = vae_resnet(d.sub("mid.block_2."), h)
    for i in reversed(range(len(ch_mult))):
        for j in range(3):
            h = vae_resnet(d.sub(f"up.{i}.block.{j}."), h)
        if i != 0:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = conv(d.sub(f"up.{i}.upsample.conv."), h, padding=1)
    h = conv(d.sub("conv_out."), F.silu(group_norm(d.sub("norm_out."), h)), padding=1)
    return ((h + 1.0) * 0.5).clamp(0.0, 1.0).numpy()  # vae.hpp:24-30


def vae_encode_moments(engine, rgb, ch_mult=(1, 2, 4, 4), num_res_blocks=2, use_quant=True):
    """VAE::encode's graph — Encoder::forward + quant_conv (auto_encoder_kl.hpp:276-366, 637-664), ldm's Encoder written from its definition: x * 2 - 1, conv_in,
    per level 2 ResnetBlocks + (pad right / bottom by one, 3x3 stride-2 conv), mid block / attention / block, GroupNorm + SiLU + conv_out -> moments (mean | logvar)."""
    w = Weights(engine, "first_stage_model.")
    e = w.sub("encoder.")
    h = conv(e.sub("conv_in."), torch.as_tensor(rgb, dtype=torch.float32) * 2.0 - 1.0, padding=1)
    for i in range(len(ch_mult)):
        for j in range(num_res_blocks):
            h = vae_resnet(e.sub(f"down.{i}.block.{j}."), h)
        if i != len(ch_mult) - 1:
            h = conv(e.sub(f"down.{i}.downsample.conv."), F.pad(h, (0, 1, 0, 1)), stride=2)
    h = vae_resnet(e.sub("mid.block_1."), h)
    h = vae_attn(e.sub("mid.attn_1."), h)
    h = vae_resnet(e.sub("mid.block_2."), h)
    h = conv(e.sub("conv_out."), F.silu(group_norm(e.sub("norm_out."), h)), padding=1)
    if use_quant:
        h = conv(w.sub("quant_conv."), h)
    return h.numpy()


# =====================================================================================================
# MMDiT (SD3 / SD3.5) — src/model/diffusion/mmdit.hpp, written from the model's mathematical definition
# =====================================================================================================
MMDIT_CFG = {
    # name: (depth, hidden, patch, in_ch, out_ch, pos_embed_max_size, d_self, qk_rms)
    "SD35_LARGE": (38, 2432, 2, 16, 16, 192, -1, True),
    "SD35_TINY": (3, 192, 2, 16, 16, 24, 0, True),
    "SD3M_TINY": (3, 192, 2, 16, 16, 24, -1, False),   # SD3-medium's variant: no qk-norm, no MMDiT-X block
}


def _rms(w: Weights, x, eps=1e-6):
    # RMSNorm over the head dim, ggml_extend.hpp:3996-4023
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w("weight")


def _ln_plain(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps=eps)


def _dit_qkv(w: Weights, x, heads, qk_rms):
    # SelfAttention::pre_attention, mmdit.hpp:326-350: fused qkv projection, per-head RMS norm of q and k
    B, L, C = x.shape
    q, k, v = linear(w.sub("qkv."), x).view(B, L, 3, C).unbind(2)
    if qk_rms:
        q = _rms(w.sub("ln_q."), q.reshape(B, L, heads, C // heads)).reshape(B, L, C)
        k = _rms(w.sub("ln_k."), k.reshape(B, L, heads, C // heads)).reshape(B, L, C)
    return q, k, v


def _modulate(x, shift, scale):
    # mmdit.hpp:368-380
    return x * (1 + scale[:, None, :]) + shift[:, None, :]


def _dit_block_pre(w: Weights, x, c, heads, qk_rms, pre_only, self_attn):
    n_mods = 9 if self_attn else (2 if pre_only else 6)
    m = linear(w.sub("adaLN_modulation.1."), F.silu(c)).chunk(n_mods, dim=-1)
    xn = _ln_plain(x)
    qkv = _dit_qkv(w.sub("attn."), _modulate(xn, m[0], m[1]), heads, qk_rms)
    qkv2 = _dit_qkv(w.sub("attn2."), _modulate(xn, m[6], m[7]), heads, qk_rms) if self_attn else None
    return qkv, qkv2, m


def _dit_block_post(w: Weights, x, attn_out, attn2_out, m, self_attn):
    # DismantledBlock::post_attention(_x), mmdit.hpp:491-556
    x = x + linear(w.sub("attn.proj."), attn_out) * m[2][:, None, :]
    if self_attn:
        x = x + linear(w.sub("attn2.proj."), attn2_out) * m[8][:, None, :]
    h = _modulate(_ln_plain(x), m[3], m[4])
    h = linear(w.sub("mlp.fc2."), F.gelu(linear(w.sub("mlp.fc1."), h), approximate="tanh"))
    return x + h * m[5][:, None, :]


def mmdit_forward(engine, model: str, x, timesteps, context, y=None):
    """x [N,16,H,W], timesteps [N], context [N,L,context_size], y [N,adm] -> [N,16,H,W]  (MMDiT::forward, mmdit.hpp:881-927)"""
    depth, hidden, ps, in_ch, out_ch, pmax, d_self, qk_rms = MMDIT_CFG[model]
    w = Weights(engine, "model.diffusion_model.")
    x = torch.from_numpy(x).float()
    t = torch.from_numpy(timesteps).float()
    context = torch.from_numpy(context).float()
    N, _, H, W = x.shape
    if context.shape[0] != N:
        context = context.repeat(N // context.shape[0], 1, 1)
    pad_h, pad_w = (ps - H % ps) % ps, (ps - W % ps) % ps
    xp = F.pad(x, (0, pad_w, 0, pad_h))
    tok = F.conv2d(xp, w("x_embedder.proj.weight"), w("x_embedder.proj.bias"), stride=ps)  # [N, hidden, h, w]
    h_, w_ = tok.shape[2], tok.shape[3]
    tok = tok.flatten(2).transpose(1, 2)  # [N, h*w, hidden]
    # cropped_pos_embed, mmdit.hpp:808-847: centre crop of the [pmax, pmax] table
    pe = w("pos_embed").reshape(pmax, pmax, hidden)
    hh, ww = (H + 1) // ps, (W + 1) // ps
    top, left = (pmax - hh) // 2, (pmax - ww) // 2
    pe = pe[top:top + hh, left:left + ww].reshape(1, hh * ww, hidden)
    xt = tok + pe
    c = linear(w.sub("t_embedder.mlp.2."), F.silu(linear(w.sub("t_embedder.mlp.0."), timestep_embedding(t, 256))))
    if y is not None:
        yy = torch.from_numpy(y).float()
        if yy.shape[0] != N:
            yy = yy.repeat(N // yy.shape[0], 1)
        c = c + linear(w.sub("y_embedder.mlp.2."), F.silu(linear(w.sub("y_embedder.mlp.0."), yy)))
    ctx = linear(w.sub("context_embedder."), context)
    heads = depth
    for i in range(depth):
        wb = w.sub(f"joint_blocks.{i}.")
        pre_only = i == depth - 1
        self_attn = i <= d_self
        cq, _, cm = _dit_block_pre(wb.sub("context_block."), ctx, c, heads, qk_rms, pre_only, False)
        xq, xq2, xm = _dit_block_pre(wb.sub("x_block."), xt, c, heads, qk_rms, False, self_attn)
        Lc = ctx.shape[1]
        joint = attention(torch.cat([cq[0], xq[0]], 1), torch.cat([cq[1], xq[1]], 1), torch.cat([cq[2], xq[2]], 1), heads)
        attn2 = attention(*xq2, heads) if self_attn else None
        new_ctx = None if pre_only else _dit_block_post(wb.sub("context_block."), ctx, joint[:, :Lc], None, cm, False)
        xt = _dit_block_post(wb.sub("x_block."), xt, joint[:, Lc:], attn2, xm, self_attn)
        ctx = new_ctx
    shift, scale = linear(w.sub("final_layer.adaLN_modulation.1."), F.silu(c)).chunk(2, dim=-1)
    out = linear(w.sub("final_layer.linear."), _modulate(_ln_plain(xt), shift, scale))  # [N, h*w, ps*ps*C] with C fastest
    out = out.view(N, h_, w_, ps, ps, out_ch).permute(0, 5, 1, 3, 2, 4).reshape(N, out_ch, h_ * ps, w_ * ps)
    return out[:, :, :H, :W].contiguous().numpy()


# =====================================================================================================
# FLUX.1 — src/model/diffusion/flux.hpp, written from the model's mathematical definition (BFL reference semantics)
# =====================================================================================================
FLUX_CFG = {
    # name: (hidden, heads, depth, single_depth, axes_dim, theta, guidance_embed)
    "FLUX_DEV": (3072, 24, 19, 38, (16, 56, 56), 10000.0, True),
    "FLUX_TINY": (128, 4, 2, 2, (8, 12, 12), 10000.0, True),
}


def flux_rope_table(h_len, w_len, n_txt, axes_dim, theta):
    """cos / sin per token and rotary pair: ids (0, row, col) for image patches, zeros for text (text first) — rope.hpp:55-106, 398-490"""
    ids = torch.zeros(n_txt + h_len * w_len, 3)
    rr, cc = torch.meshgrid(torch.arange(h_len, dtype=torch.float32), torch.arange(w_len, dtype=torch.float32), indexing="ij")
    ids[n_txt:, 1] = rr.reshape(-1)
    ids[n_txt:, 2] = cc.reshape(-1)
    cos, sin = [], []
    for a, dim in enumerate(axes_dim):
        half = dim // 2
        scale = torch.linspace(0.0, (dim - 2) / dim, half) if half > 1 else torch.zeros(1)
        omega = 1.0 / (theta ** scale)
        ang = ids[:, a:a + 1] * omega[None]
        cos.append(torch.cos(ang))
        sin.append(torch.sin(ang))
    return torch.cat(cos, -1), torch.cat(sin, -1)  # [L, d_head/2]


def _apply_rope(x, cos, sin):
    # x [B, heads, L, d]; interleaved pairs (x0, x1) -> (x0 cos - x1 sin, x0 sin + x1 cos)  (rope.hpp:966-1004)
    x0, x1 = x[..., 0::2], x[..., 1::2]
    out = torch.stack([x0 * cos - x1 * sin, x0 * sin + x1 * cos], dim=-1)
    return out.flatten(-2)


def _flux_qkv(qkv, heads):
    B, L, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.split(C, dim=-1)
    return [t.view(B, L, heads, C // heads).transpose(1, 2) for t in (q, k, v)]  # [B, heads, L, d]


def _flux_rms(w: Weights, x, eps=1e-6):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w("scale")


def _flux_attn(q, k, v, cos, sin):
    q, k = _apply_rope(q, cos, sin), _apply_rope(k, cos, sin)
    o = F.scaled_dot_product_attention(q, k, v)
    B, Hh, L, d = o.shape
    return o.transpose(1, 2).reshape(B, L, Hh * d)


def flux_forward(engine, model: str, x, timesteps, context, y, guidance=3.5):
    """x [N,16,H,W], timesteps [N] (sigma in (0,1]), context [N|1,L,ctx], y [N|1,vec] -> [N,16,H,W]  (flux.hpp:1008-1337)"""
    hidden, heads, depth, sdepth, axes, theta, gemb = FLUX_CFG[model]
    w = Weights(engine, "model.diffusion_model.")
    x = torch.from_numpy(x).float()
    N, C, H, W = x.shape
    t = torch.from_numpy(timesteps).float()
    ctx = torch.from_numpy(context).float()
    yy = torch.from_numpy(y).float()
    if ctx.shape[0] != N:
        ctx = ctx.repeat(N // ctx.shape[0], 1, 1)
    if yy.shape[0] != N:
        yy = yy.repeat(N // yy.shape[0], 1)
    ps = 2
    xp = F.pad(x, (0, (ps - W % ps) % ps, 0, (ps - H % ps) % ps))
    hh, ww = xp.shape[2] // ps, xp.shape[3] // ps
    # patchify, channel-major then (ph, pw) ("patch_last"): [N, h*w, C*ps*ps]
    img = xp.view(N, C, hh, ps, ww, ps).permute(0, 2, 4, 1, 3, 5).reshape(N, hh * ww, C * ps * ps)
    img = linear(w.sub("img_in."), img)

    def embed(wm, v):
        return linear(wm.sub("out_layer."), F.silu(linear(wm.sub("in_layer."), v)))

    vec = embed(w.sub("time_in."), timestep_embedding(t * 1000.0, 256))
    if gemb:
        vec = vec + embed(w.sub("guidance_in."), timestep_embedding(torch.full((N,), float(guidance)) * 1000.0, 256))
    vec = vec + embed(w.sub("vector_in."), yy)
    txt = linear(w.sub("txt_in."), ctx)
    n_txt = txt.shape[1]
    cos, sin = flux_rope_table(hh, ww, n_txt, axes, theta)

    def mods(wm, n):
        return linear(wm.sub("lin."), F.silu(vec)).chunk(n, dim=-1)

    def mlp(wm, v):
        return linear(wm.sub("2."), F.gelu(linear(wm.sub("0."), v), approximate="tanh"))

    for i in range(depth):
        wb = w.sub(f"double_blocks.{i}.")
        im, tm = mods(wb.sub("img_mod."), 6), mods(wb.sub("txt_mod."), 6)
        iq = _flux_qkv(linear(wb.sub("img_attn.qkv."), _modulate(_ln_plain(img), im[0], im[1])), heads)
        tq = _flux_qkv(linear(wb.sub("txt_attn.qkv."), _modulate(_ln_plain(txt), tm[0], tm[1])), heads)
        iq[0], iq[1] = _flux_rms(wb.sub("img_attn.norm.query_norm."), iq[0]), _flux_rms(wb.sub("img_attn.norm.key_norm."), iq[1])
        tq[0], tq[1] = _flux_rms(wb.sub("txt_attn.norm.query_norm."), tq[0]), _flux_rms(wb.sub("txt_attn.norm.key_norm."), tq[1])
        attn = _flux_attn(*(torch.cat([a, b], dim=2) for a, b in zip(tq, iq)), cos, sin)
        ta, ia = attn[:, :n_txt], attn[:, n_txt:]
        img = img + linear(wb.sub("img_attn.proj."), ia) * im[2][:, None]
        img = img + mlp(wb.sub("img_mlp."), _modulate(_ln_plain(img), im[3], im[4])) * im[5][:, None]
        txt = txt + linear(wb.sub("txt_attn.proj."), ta) * tm[2][:, None]
        txt = txt + mlp(wb.sub("txt_mlp."), _modulate(_ln_plain(txt), tm[3], tm[4])) * tm[5][:, None]
    xs = torch.cat([txt, img], dim=1)
    for i in range(sdepth):
        wb = w.sub(f"single_blocks.{i}.")
        m = mods(wb.sub("modulation."), 3)
        l1 = linear(wb.sub("linear1."), _modulate(_ln_plain(xs), m[0], m[1]))
        q, k, v = _flux_qkv(l1[..., :3 * hidden], heads)
        q, k = _flux_rms(wb.sub("norm.query_norm."), q), _flux_rms(wb.sub("norm.key_norm."), k)
        attn = _flux_attn(q, k, v, cos, sin)
        out = linear(wb.sub("linear2."), torch.cat([attn, F.gelu(l1[..., 3 * hidden:], approximate="tanh")], dim=-1))
        xs = xs + out * m[2][:, None]
    img = xs[:, n_txt:]
    shift, scale = linear(w.sub("final_layer.adaLN_modulation.1."), F.silu(vec)).chunk(2, dim=-1)
    out = linear(w.sub("final_layer.linear."), _modulate(_ln_plain(img), shift, scale))  # [N, h*w, C*ps*ps]
    out = out.view(N, hh, ww, C, ps, ps).permute(0, 3, 1, 4, 2, 5).reshape(N, C, hh * ps, ww * ps)
    return out[:, :, :H, :W].contiguous().numpy()
