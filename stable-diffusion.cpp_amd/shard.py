"""Image sharding across the GPUs of one node.

Each image (seed + b) is an independent denoise trajectory, so ranks never exchange data on the hot path: rank r owns
images {b : b % world == r}.  The only collective is the optional gather of finished results to rank 0
(torch.distributed: backend "nccl" == RCCL over xGMI on MI355X, "gloo" in the CPU tests) — payload <= a few MB."""
from __future__ import annotations

import numpy as np


def shard_indices(batch_count: int, rank: int, world: int) -> list[int]:
    return [b for b in range(batch_count) if b % world == rank]


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
