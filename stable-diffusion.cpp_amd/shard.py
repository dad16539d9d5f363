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


class _DeviceF32:
    """A flat f32 device buffer exposed through __cuda_array_interface__ (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}


def make_pair_exchange(dist, group=None):
    """The per-step exchange of the CFG-pair split as a torch.distributed all-reduce (SUM) on the engine's own eps buffer.

    backend "nccl" (= RCCL on ROCm): the buffer is wrapped as a CUDA tensor in place and the collective is ordered on the engine's HIP stream
    (torch.cuda.ExternalStream) — no host copy, no host synchronisation; one [N,C,H,W] f32 all-reduce per step (64 KB per SD1.5 image) over a
    single xGMI link.  backend "gloo" (CPU tests, host backends): the buffer is host memory and is reduced in place."""
    import ctypes

    import torch

    def exchange(ptr: int, count: int, stream) -> bool:
        # The MI355X engine hands over a DEVICE address together with its (non-null) stream; host backends (the CPU oracle in the gloo tests)
        # pass host memory and no stream.  The branch follows the pointer, not the process group's backend (a gloo group with GPU
        # engines used to read the device pointer as host memory).
        if stream:
            dev = torch.device("cuda", torch.cuda.current_device())
            t = torch.as_tensor(_DeviceF32(ptr, count), device=dev)
            ext = torch.cuda.ExternalStream(int(stream), device=dev)
            if dist.get_backend(group) == "nccl":
                with torch.cuda.stream(ext):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:  # no device collective in this group: stage through host memory, ordered on the engine's stream
                with torch.cuda.stream(ext):
                    h = t.to("cpu", non_blocking=False)
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                    t.copy_(h)
                ext.synchronize()
        elif dist.get_backend(group) == "nccl":
            raise RuntimeError("pair exchange: a host buffer cannot be reduced over an nccl (RCCL) group")
        else:
            a = np.ctypeslib.as_array((ctypes.c_float * count).from_address(ptr))
            dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM, group=group)
        return True

    return exchange


def sample_cfg_pair_split(engine, cond, uncond, *, width: int, height: int, steps: int, cfg: float, seed: int, dist, group=None, rank_in_pair: int,
                          batch: int = 1, eta: float = float("inf"), ancestral: bool = True, cond_y=None, uncond_y=None, exchange=None) -> np.ndarray:
    """CFG-pair split (the one real exchange step on this path): when there are fewer images than GPUs, the cond
    branch runs on one rank of a pair and the uncond branch on the other; per step the pair all-reduces (SUM) its PRE-SCALED eps —
    s*cond on the cond rank, (1-s)*uncond on the other — so the sum is uncond + s*(cond - uncond) (src/runtime/guidance.cpp:171).
    The whole trajectory stays in the engine and in HBM (sd_set_pair_exchange + the device-resident sampler): the engine hands the exchange the
    device address of its eps buffer and the stream it was produced on, `make_pair_exchange` runs the all-reduce there.  Both ranks then take
    the same Euler(-A) update (src/runtime/denoiser.hpp:1513-1546, 1582-1597) with the same Philox noise, so their latents stay bit-identical
    and nothing else is exchanged.

    rank_in_pair: 0 = cond branch, 1 = uncond branch.  Returns the final latents [batch, C, H/8, W/8] (identical on both ranks)."""
    from . import EULER, EULER_A

    engine.set_pair_exchange(exchange or make_pair_exchange(dist, group), rank_in_pair)
    try:
        return engine.sample_latents(cond, uncond, width=width, height=height, steps=steps, cfg=cfg, seed=seed, batch=batch, device_batch=batch,
                                     method=EULER_A if ancestral else EULER, eta=eta, cond_y=cond_y, uncond_y=uncond_y, device_sampler=True)
    finally:
        engine.set_pair_exchange(None)


def generate_multi_device(engines, cond, uncond, *, batch_count: int, seed: int, decode=False, **kw) -> dict:
    """One process x N devices: `engines` are contexts created on different devices (Engine(backend="MI355X<i>")); every engine runs its
    contiguous share of the images (shard_indices) on its own host thread — each backend instance owns its device stream, plan cache and
    arena, and ctypes releases the GIL for the duration of a call, so the N trajectories run concurrently.  The alternative to one process per
    GPU (torchrun) when the caller is a single server process.  Returns {image index: result}."""
    import threading

    world = len(engines)
    results, errors = [None] * world, [None] * world

    def run(r):
        try:
            results[r] = generate_sharded(engines[r], cond, uncond, batch_count=batch_count, seed=seed, rank=r, world=world, decode=decode, **kw)
        except BaseException as e:  # surfaced on the caller's thread
            errors[r] = e

    threads = [threading.Thread(target=run, args=(r,), name=f"sd-device-{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    out = {}
    for r in results:
        out.update(r)
    return out
