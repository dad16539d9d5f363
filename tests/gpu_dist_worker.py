"""Worker process of tests/test_gpu_dist.py (one scenario per invocation: argv[1]).  torch's HIP runtime is initialised BEFORE the backend
plug-in is loaded — the torch wheel bundles its own libamdhip64 and cannot initialise once /opt/rocm's copy is resident (bench.py has the
same order)."""
import ctypes
import os
import socket
import sys
import threading
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
ON_GPU = not os.environ.get("SDCPP_GPU_TESTS_ON_ORACLE")
import torch

if ON_GPU:
    torch.cuda.init()
    torch.cuda.set_device(0)
import sdcpp_amd as sd
from sdcpp_amd import shard

if ON_GPU:
    sd.load_mi355x_backend()
    GPU = "MI355X0"
    for kv in filter(None, os.environ.get("SDCPP_BACKEND_OPTS", "").split(",")):  # debugging aid, as in conftest.py
        sd.backend_set_option(kv.split("=")[0].strip(), int(kv.split("=")[1]))
else:
    sd.load_backend(ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so")
    GPU = "CPU-oracle"


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def conds(seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((1, 77, 64)).astype(np.float32), rng.standard_normal((1, 77, 64)).astype(np.float32)


KW = dict(width=64, height=64, steps=3, cfg=7.0, seed=77)


def pair_split(ancestral):
    cond, uncond = conds(12)
    engines = [sd.Engine(model=sd.SD15_TINY, backend=GPU) for _ in range(2)]
    bar = threading.Barrier(2)
    slots = [None, None]

    def make(rank):
        def exchange(ptr, count, stream):
            if ON_GPU:
                assert stream, "the MI355X backend must hand over its stream"
                torch.cuda.ExternalStream(int(stream), device=torch.device("cuda", 0)).synchronize()
                slots[rank] = torch.as_tensor(shard._DeviceF32(ptr, count), device="cuda:0")
            else:
                slots[rank] = torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * count).from_address(ptr)))
            bar.wait()
            if rank == 0:
                total = slots[0] + slots[1]
                slots[0].copy_(total)
                slots[1].copy_(total)
                if ON_GPU:
                    torch.cuda.synchronize()
            bar.wait()
            return True
        return exchange

    out, err = [None, None], [None, None]

    def run(rank):
        try:
            out[rank] = shard.sample_cfg_pair_split(engines[rank], cond, uncond, dist=None, rank_in_pair=rank, batch=2, ancestral=ancestral,
                                                    exchange=make(rank), **KW)
        except BaseException as e:
            err[rank] = e
            bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert err == [None, None], err
    np.testing.assert_array_equal(out[0], out[1])
    ref = engines[0].sample_latents(cond, uncond, batch=2, device_batch=2, method=sd.EULER_A if ancestral else sd.EULER, device_sampler=True, **KW)
    e = rel_l2(out[0], ref)
    print(f"pair split (ancestral={ancestral}) vs single-context CFG trajectory: rel-L2 {e:.2e}")
    assert np.isfinite(out[0]).all() and e < 2e-3   # two branch graphs of batch 2 vs one fused graph of batch 4: tile choices differ


def rccl_in_place():
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        cond, uncond = conds(13)
        eng = sd.Engine(model=sd.SD15_TINY, backend=GPU)
        calls = []

        def noop(ptr, count, stream):
            calls.append(count)
            return True

        a = shard.sample_cfg_pair_split(eng, cond, uncond, dist=None, rank_in_pair=0, batch=1, exchange=noop, **KW)
        b = shard.sample_cfg_pair_split(eng, cond, uncond, dist=dist, rank_in_pair=0, batch=1, **KW)
        assert calls == [4 * 8 * 8] * KW["steps"], calls
        np.testing.assert_array_equal(a, b)
        assert np.isfinite(a).all() and float(np.abs(a).max()) > 0
    finally:
        dist.destroy_process_group()


def rccl_native():
    """The C++ exchange (csrc/host/rccl_exchange.cpp: dlopen'ed librccl, ncclAllReduce on the backend stream, no torch.distributed) on a one-rank
    communicator: the all-reduce is the identity, so the trajectory equals the one with a no-op exchange bit for bit — and the call count /
    buffer are the engine's own."""
    cond, uncond = conds(13)
    eng = sd.Engine(model=sd.SD15_TINY, backend=GPU)

    def noop(ptr, count, stream):
        return True

    a = shard.sample_cfg_pair_split(eng, cond, uncond, dist=None, rank_in_pair=0, batch=1, exchange=noop, **KW)
    comm = sd.rccl_comm_create(0, 1, 0, sd.rccl_unique_id())
    try:
        eng.set_pair_exchange_rccl(comm, 0)
        b = eng.sample_latents(cond, uncond, batch=1, device_batch=1, method=sd.EULER_A, device_sampler=True, **KW)
    finally:
        eng.set_pair_exchange_rccl(None)
        sd.rccl_comm_destroy(comm)
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).all() and float(np.abs(a).max()) > 0


def multi_device():
    cond, uncond = conds(14)
    engines = [sd.Engine(model=sd.SD15_TINY, backend=GPU) for _ in range(2)]
    res = shard.generate_multi_device(engines, cond, uncond, batch_count=5, width=64, height=64, steps=2, cfg=7.0, seed=100, device_sampler=True)
    assert sorted(res) == [0, 1, 2, 3, 4]
    ref = engines[0].sample_latents(cond, uncond, width=64, height=64, steps=2, cfg=7.0, seed=100, batch=5, device_batch=5, device_sampler=True)
    for b in range(5):
        assert rel_l2(res[b], ref[b]) < 2e-3


{"pair_a": lambda: pair_split(True), "pair_e": lambda: pair_split(False), "rccl": rccl_in_place, "rccl_native": rccl_native,
 "multi": multi_device}[sys.argv[1]]()
print("OK")
