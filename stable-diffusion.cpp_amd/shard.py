"""Image sharding across the GPUs of one node.

Each image (seed + b) is an independent denoise trajectory, so ranks never exchange data on the hot path: rank r owns the CONTIGUOUS
block of images [r * ceil(B / world), (r + 1) * ceil(B / world)) — contiguous seeds are what the engine batches into one device graph
(seeds seed+b0 .. seed+b0+n-1), so every rank runs its whole share as ONE device batch (a round-robin b % world assignment would leave
each rank with n sequential batch-1 trajectories).  The only collective is the optional gather of finished results
(torch.distributed: backend "nccl" == RCCL over xGMI on MI355X, "gloo" in the CPU tests) — payload <= a few MB."""
from __future__ import annotations

import numpy as np


def shard_indices(batch_count: int, rank: int, world: int) -> list[int]:
    per = -(-batch_count // world)
    return list(range(min(rank * per, batch_count), min((rank + 1) * per, batch_count)))


def generate_sharded(engine, cond, uncond, *, batch_count: int, seed: int, rank: int, world: int, gather=None, decode=False, **kw):
    """Run this rank's share (seeds seed+b for its b's, as one device batch); optionally gather to every rank.

    `gather` is a callable (local_array, list_of_indices) -> dict{index: array} built by the caller from torch.distributed
    (see tests/test_dist_shard.py / bench.py); with gather=None only the local results are returned."""
    mine = shard_indices(batch_count, rank, world)
    out = {}
    fn = engine.generate_image if decode else engine.sample_latents
    # contiguous seed runs share one device batch
    runs, cur = [], []
    for b in mine:
        if cur and b != cur[-1] + 1:
            runs.append(cur)
            cur = []
        cur.append(b)
    if cur:
        runs.append(cur)
    for r in runs:
        res = fn(cond, uncond, seed=seed + r[0], batch=len(r), device_batch=len(r), **kw)
        for i, b in enumerate(r):
            out[b] = res[i]
    if gather is not None:
        out = gather(out)
    return out


def sample_cfg_pair_split(engine, cond, uncond, *, width: int, height: int, steps: int, cfg: float, seed: int, dist, group=None, rank_in_pair: int,
                          batch: int = 1, eta: float = 1.0, ancestral: bool = True, cond_y=None, uncond_y=None) -> np.ndarray:
    """CFG-pair split (the one real exchange step on this path): when there are fewer images than GPUs, the cond
    branch runs on one rank of a pair and the uncond branch on the other; per step the pair all-reduces (SUM) its PRE-SCALED eps —
    s*cond on the cond rank, (1-s)*uncond on the other — so the sum is uncond + s*(cond - uncond) (src/runtime/guidance.cpp:171).
    One [N,C,H,W] f32 all-reduce per step (64 KB per SD1.5 image) over a single xGMI link with RCCL (`dist` = torch.distributed with
    backend "nccl"; "gloo" in the CPU tests).  Both ranks then take the same Euler(-A) update (src/runtime/denoiser.hpp:1513-1546,
    1582-1597) with the same Philox noise, so their latents stay bit-identical and nothing else is exchanged.

    rank_in_pair: 0 = cond branch, 1 = uncond branch.  Returns the final latents [batch, C, H/8, W/8] (identical on both ranks)."""
    import torch

    from . import get_sigmas, lib, philox_randn

    h, w = height // 8, width // 8
    C = 4
    per = C * h * w
    sig = get_sigmas(steps)
    x = np.stack([philox_randn(seed + b, 0, per).reshape(C, h, w) * sig[0] for b in range(batch)]).astype(np.float32)
    offs = [1] * batch
    mine, my_y = (cond, cond_y) if rank_in_pair == 0 else (uncond, uncond_y)
    weight = np.float32(cfg) if rank_in_pair == 0 else np.float32(1.0 - cfg)
    for i in range(steps):
        s, s_to = np.float32(sig[i]), np.float32(sig[i + 1])
        c_in = np.float32(1.0) / np.sqrt(s * s + np.float32(1.0))   # CompVisDenoiser::get_scalings, denoiser.hpp:1167-1172
        t = np.full((batch,), lib().sd_sigma_to_t(float(s)), dtype=np.float32)
        eps = engine.unet_forward(x * c_in, t, mine, my_y) * weight
        buf = torch.from_numpy(np.ascontiguousarray(eps))
        if dist.get_backend(group) == "nccl":
            dev = buf.cuda()
            dist.all_reduce(dev, op=dist.ReduceOp.SUM, group=group)
            guided = dev.cpu().numpy()
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            guided = buf.numpy()
        den = guided * (-s) + x
        if not ancestral or s_to == 0:
            x = den if s_to == 0 else x + (x - den) / s * (s_to - s)
            continue
        up = np.float32(min(float(s_to), eta * float(np.sqrt(max(float(s_to) ** 2 * (float(s) ** 2 - float(s_to) ** 2) / float(s) ** 2, 0.0)))))
        down = np.float32(np.sqrt(max(float(s_to) ** 2 - float(up) ** 2, 0.0)))
        r = np.float32(down / s)
        x = r * x + (np.float32(1) - r) * den
        for b in range(batch):
            x[b] = x[b] + philox_randn(seed + b, offs[b], per).reshape(C, h, w) * up
            offs[b] += 1
    return x.astype(np.float32)
