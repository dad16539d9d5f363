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
