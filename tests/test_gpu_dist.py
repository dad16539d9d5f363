"""The multi-GPU plumbing on ONE GPU (the GPU box has a single device; the 2-rank path itself is covered over gloo in test_dist_shard.py):
  * the CFG-pair split's exchange hook works on the engine's DEVICE buffer and stream — two engine contexts on the same device play the two
    ranks from two host threads, the "collective" is a device-side add;
  * the RCCL call of shard.make_pair_exchange runs on that buffer in place (process group of one rank over 127.0.0.1);
  * one process x N backend instances: shard.generate_multi_device on two contexts.
Each scenario runs in its own process (tests/gpu_dist_worker.py): torch's HIP runtime has to be initialised before the backend plug-in."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(scenario):
    env = dict(os.environ, OMP_NUM_THREADS="4", OMP_WAIT_POLICY="PASSIVE")
    cmd = [sys.executable, str(ROOT / "tests" / "gpu_dist_worker.py"), scenario]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    # (Until round 6 a worker killed by a signal was retried once: a "Memory access fault by GPU node" had been seen once in ~40 runs.  The mechanism found
    # and removed in round 6: the conv kernels' zero page was a thread-local, lazily allocated inside the launch function — when a plan's first use of it
    # on some thread fell inside a hipGraph capture the allocation failed, the page stayed NULL and the padding taps read address 0.  It is now one
    # process-wide page per device made at backend creation.  No retry: scripts/gpu_pair_loop.sh ran this scenario 200 times clean, profiles/r07*_pair_loop.txt.)
    # (RCCL prints its version banner through C stdio at exit, i.e. after the worker's last line)
    assert p.returncode == 0 and "OK" in p.stdout.split(), f"{scenario}:\n{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    return p.stdout


@pytest.mark.parametrize("scenario", ["pair_a", "pair_e"])
def test_cfg_pair_split_device_side_exchange(sd, gpu, scenario):
    """cond branch on one context, uncond on the other, one in-place sum of the two device buffers per step (Euler-A / Euler): both contexts
    end with the same latents, equal to the single-context CFG trajectory."""
    print(_run(scenario))


def test_cfg_pair_exchange_is_an_in_place_rccl_all_reduce(sd, gpu):
    """shard.make_pair_exchange on a one-rank process group: the all-reduce must leave the engine's buffer as it was (sum over one rank), i.e.
    the trajectory equals the one with a no-op exchange — the collective really ran on the device buffer the engine reads next."""
    if os.environ.get("SDCPP_GPU_TESTS_ON_ORACLE"):
        pytest.skip("RCCL needs the GPU")
    _run("rccl")


def test_cfg_pair_exchange_native_rccl_from_cpp(sd, gpu):
    """sd_set_pair_exchange_rccl: the exchange issued from C++ (librccl.so loaded with dlopen next to the plug-in's HIP runtime, ncclAllReduce on the
    backend stream; replaces guidance.cpp:149-179 when cond / uncond sit on two GPUs) — one-rank communicator, bit-identical to a no-op exchange."""
    if os.environ.get("SDCPP_GPU_TESTS_ON_ORACLE"):
        pytest.skip("RCCL needs the GPU")
    _run("rccl_native")


def test_one_process_two_backend_instances(sd, gpu):
    """shard.generate_multi_device: two contexts, each running its contiguous share of five images on its own thread, equal the single-context
    batch image by image."""
    _run("multi")
