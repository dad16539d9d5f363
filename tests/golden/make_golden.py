"""Generates tests/golden/ops_torch_fp32.npz: seeded inputs + PyTorch-CPU fp32 outputs for every ggml op on the hot path
(SURVEY.md section 2.3).  PyTorch is the INDEPENDENT mathematical reference the CPU oracle is pinned against — the
reference repo itself ships no golden vectors or tests for this path and its ggml submodule is absent.

    python tests/golden/make_golden.py          # rewrites the .npz (deterministic: numpy default_rng(1234))
"""
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

OUT = Path(__file__).resolve().parent / "ops_torch_fp32.npz"


def main():
    rng = np.random.default_rng(1234)
    T = lambda a: torch.from_numpy(a)
    d = {}
    # MUL_MAT (Linear): x [tokens,K], w [M,K] stored f16
    x = rng.standard_normal((37, 96)).astype(np.float32)
    w = (rng.standard_normal((50, 96)) / np.sqrt(96)).astype(np.float32)
    b = rng.standard_normal(50).astype(np.float32)
    w16 = w.astype(np.float16).astype(np.float32)
    d.update(lin_x=x, lin_w=w, lin_b=b, lin_y=F.linear(T(x), T(w16), T(b)).numpy())
    # conv 3x3 s1 p1, 3x3 s2 p1, 1x1
    cx = rng.standard_normal((2, 12, 9, 11)).astype(np.float32)
    cw = (rng.standard_normal((20, 12, 3, 3)) / np.sqrt(108)).astype(np.float32)
    cb = rng.standard_normal(20).astype(np.float32)
    cw16 = cw.astype(np.float16).astype(np.float32)
    d.update(conv_x=cx, conv_w=cw, conv_b=cb,
             conv_y_s1=F.conv2d(T(cx), T(cw16), T(cb), stride=1, padding=1).numpy(),
             conv_y_s2=F.conv2d(T(cx), T(cw16), T(cb), stride=2, padding=1).numpy())
    pw = (rng.standard_normal((7, 12, 1, 1)) / np.sqrt(12)).astype(np.float32)
    d.update(conv1_w=pw, conv1_y=F.conv2d(T(cx), T(pw.astype(np.float16).astype(np.float32))).numpy())
    # group norm (32 groups, eps 1e-6) + affine + silu ; layer norm eps 1e-5 ; rms norm
    gx = (rng.standard_normal((2, 64, 6, 5)) * 2 + 0.3).astype(np.float32)
    gw = rng.standard_normal(64).astype(np.float32)
    gb = rng.standard_normal(64).astype(np.float32)
    d.update(gn_x=gx, gn_w=gw, gn_b=gb, gn_y=F.silu(F.group_norm(T(gx), 32, T(gw), T(gb), eps=1e-6)).numpy())
    lx = (rng.standard_normal((11, 80)) * 3 - 1).astype(np.float32)
    lw = rng.standard_normal(80).astype(np.float32)
    lb = rng.standard_normal(80).astype(np.float32)
    d.update(ln_x=lx, ln_w=lw, ln_b=lb, ln_y=F.layer_norm(T(lx), (80,), T(lw), T(lb), eps=1e-5).numpy(),
             rms_y=(T(lx) * torch.rsqrt(T(lx).pow(2).mean(-1, keepdim=True) + 1e-6)).numpy())
    # activations
    ax = (rng.standard_normal((5, 64)) * 3).astype(np.float32)
    d.update(act_x=ax, silu_y=F.silu(T(ax)).numpy(), gelu_y=F.gelu(T(ax), approximate="tanh").numpy(),
             gelu_quick_y=(T(ax) * torch.sigmoid(1.702 * T(ax))).numpy(), sigmoid_y=torch.sigmoid(T(ax)).numpy())
    # softmax rows
    sx = (rng.standard_normal((3, 7, 33)) * 4).astype(np.float32)
    d.update(sm_x=sx, sm_y=F.softmax(T(sx), dim=-1).numpy())
    # attention: q [HN,Lq,d] k,v [HN,Lk,d]
    q = rng.standard_normal((4, 19, 24)).astype(np.float32)
    k = rng.standard_normal((4, 13, 24)).astype(np.float32)
    v = rng.standard_normal((4, 13, 24)).astype(np.float32)
    d.update(att_q=q, att_k=k, att_v=v, att_y=F.scaled_dot_product_attention(T(q), T(k), T(v)).numpy())
    # nearest upscale x2, concat, timestep embedding (cos first)
    ux = rng.standard_normal((1, 3, 4, 5)).astype(np.float32)
    d.update(up_x=ux, up_y=F.interpolate(T(ux), scale_factor=2, mode="nearest").numpy())
    ts = np.array([999.0, 500.5, 3.0], dtype=np.float32)
    half = 160
    freqs = np.exp(-np.log(10000.0) * np.arange(half, dtype=np.float64) / half)
    args = ts[:, None].astype(np.float64) * freqs[None]
    d.update(te_t=ts, te_y=np.concatenate([np.cos(args), np.sin(args)], -1).astype(np.float32))
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, sum(v.nbytes for v in d.values()), "bytes raw")


if __name__ == "__main__":
    main()
