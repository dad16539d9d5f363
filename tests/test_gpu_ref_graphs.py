"""Reference-emitted graphs through the MI355X backend (VERDICT r5 next-round task 2 b/c; SURVEY.md section 8 row f1, offline half).

The graphs here are built by the reference's OWN code — UnetModelBlock / AutoEncoderKLModel / MMDiT / Flux forward through the reference's build_graph — and
submitted by the reference's OWN GGMLRunner::compute (its gallocr placement, its input uploads, ggml_backend_graph_compute through
sd_backend_graph_compute_with_eval_callback, its read-back), all compiled from /root/reference into oracle/_ref/libref_graphs.so against this repository's
ggml front-end (tests/ref_graphs.py).  libggml-mi355x.so sees exactly what it would see under the real host, minus the real libggml.

  * result vs the same reference runner on the CPU oracle backend: the parity bar of the whole-graph tests;
  * result vs the engine (csrc/host/models.hpp graphs) on the GPU: BIT-IDENTICAL — the graphs are equal node for node (tests/test_ref_graphs.py), so the
    planner builds the same plan;
  * every fusion counter of ggml_backend_mi355x_get_stats advances by the same amount for the reference-emitted graph as for the engine's: the fusion
    patterns — and therefore bench.py's numbers — carry over to the real host;
  * the reference's REAL eval-callback slicing (sd_set_backend_eval_callback -> sd_ggml_graph_view, src/core/ggml_extend_backend.cpp:449-509) on the GPU:
    cut behind every MUL_MAT, bit-identical to the host restatement's sliced run (tests/test_gpu_graph_views.py).
"""
import numpy as np
import pytest

import ref_graphs as rg
from test_ref_graphs import TINY, inputs_for

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not rg.available(), reason="oracle/_ref/libref_graphs.so not built (needs /root/reference: `make -C oracle ref`)")]

NOT_FUSION_COUNTERS = {"graphs_computed", "plans_built", "graph_replays", "swizzled_weight_bytes", "view_graphs", "view_external_nodes"}


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _delta(sd, fn, on_gpu=True):
    if not on_gpu:
        return fn(), {}
    s0 = sd.backend_stats()
    out = fn()
    s1 = sd.backend_stats()
    return out, {k: s1[k] - s0[k] for k in s1 if k not in NOT_FUSION_COUNTERS}


@pytest.mark.parametrize("name", TINY)
def test_reference_emitted_graph_on_the_gpu(sd, oracle, gpu, name):
    c = inputs_for(sd, name, np.random.default_rng(11))
    post = c.get("post", lambda v: v)
    e_gpu = sd.Engine(model=c["model"], backend=gpu, flash_attn=True)
    c.get("prepare", lambda e_: None)(e_gpu)
    r_gpu = rg.RefRunner(e_gpu, c["family"], c["version"], gpu, flash_attn=True, overrides=c["overrides"])
    r_cpu = rg.RefRunner(e_gpu, c["family"], c["version"], oracle, flash_attn=True, overrides=c["overrides"])
    for r in (r_gpu, r_cpu):
        if c.get("scale"):
            r.set_conv2d_scale(c["scale"])
    on_gpu = gpu != oracle
    if on_gpu:
        ref_gpu, d_ref = _delta(sd, lambda: post(r_gpu.compute(c["out"], **c["ref"])))
        eng_gpu, d_eng = _delta(sd, lambda: c["eng"](e_gpu))
    else:
        ref_gpu, eng_gpu = post(r_gpu.compute(c["out"], **c["ref"])), c["eng"](e_gpu)
    ref_cpu = post(r_cpu.compute(c["out"], **c["ref"]))
    assert np.isfinite(ref_gpu).all()
    err = rel_l2(ref_gpu, ref_cpu)
    print(f"{name}: reference runner on {gpu} vs on the oracle: rel-L2 {err:.2e}; vs the engine on the GPU identical: {np.array_equal(ref_gpu, eng_gpu.reshape(ref_gpu.shape))}")
    assert err < (2e-2 if name == "FLUX_TINY" else 5e-3)
    if on_gpu:
        assert d_ref == d_eng, {k: (d_ref[k], d_eng[k]) for k in d_ref if d_ref[k] != d_eng[k]}
    np.testing.assert_array_equal(ref_gpu, eng_gpu.reshape(ref_gpu.shape))
    if on_gpu:
        assert d_ref["kernels_planned"] > 0 and d_ref["nodes_seen"] > 0
        print(f"{name}: fusion counters identical:", {k: v for k, v in d_ref.items() if v})
    for r in (r_gpu, r_cpu):
        r.close()


def test_full_width_sd15_unet_through_the_reference_runner_on_the_gpu(sd, oracle, gpu):
    """The benchmarked SD1.5 UNet (320 channels, 64x64 latent, batch 2) submitted by the reference's runner: bit-identical to the engine's forward."""
    c = inputs_for(sd, "SD15", np.random.default_rng(12), real=True)
    x = np.random.default_rng(13).standard_normal((2, 4, 64, 64)).astype(np.float32)
    t = np.array([700.0, 200.0], np.float32)
    ctx = c["ref"]["ctx"]
    e = sd.Engine(model=sd.SD15, backend=gpu, flash_attn=True)
    r = rg.RefRunner(e, "unet", "sd1", gpu, flash_attn=True)
    ref, d_ref = _delta(sd, lambda: r.compute(x.shape, x=x, t=t, ctx=ctx), gpu != oracle)
    out, d_eng = _delta(sd, lambda: e.unet_forward(x, t, ctx), gpu != oracle)
    assert np.isfinite(ref).all()
    np.testing.assert_array_equal(ref, out)
    if gpu != oracle:
        assert d_ref == d_eng, {k: (d_ref[k], d_eng[k]) for k in d_ref if d_ref[k] != d_eng[k]}
        print("SD1.5 UNet through the reference runner: fusion counters", {k: v for k, v in d_ref.items() if v})
    r.close()


def test_full_width_sdxl_unet_q8_0_through_the_reference_runner_on_the_gpu(sd, oracle, gpu):
    """BASELINE.json config 3's model — the SDXL UNet, 70 transformer blocks, q8_0 Linear weights + f16 conv kernels, 1024x1024 (128x128 latent), cond + uncond — built,
    placed and submitted by the reference's own runner: bit-identical to the engine's forward, same fusion counters (the raw-block kernels, sibling / hoisted
    launches and split-K decisions the bench numbers rest on are taken for the reference-emitted graph too)."""
    rng = np.random.default_rng(15)
    x = np.repeat(rng.standard_normal((1, 4, 128, 128)).astype(np.float32), 2, axis=0)
    t = np.array([600.0, 600.0], np.float32)
    ctx = rng.standard_normal((2, 77, 2048)).astype(np.float32)
    y = rng.standard_normal((2, 2816)).astype(np.float32)
    e = sd.Engine(model=sd.SDXL, backend=gpu, flash_attn=True, wtype=sd.Q8_0)
    r = rg.RefRunner(e, "unet", "sdxl", gpu, flash_attn=True)
    ref, d_ref = _delta(sd, lambda: r.compute(x.shape, x=x, t=t, ctx=ctx, y=y), gpu != oracle)
    out, d_eng = _delta(sd, lambda: e.unet_forward(x, t, ctx, y), gpu != oracle)
    assert np.isfinite(ref).all()
    if gpu != oracle:
        assert d_ref == d_eng, {k: (d_ref[k], d_eng[k]) for k in d_ref if d_ref[k] != d_eng[k]}
        print("SDXL UNet q8_0 through the reference runner: fusion counters", {k: v for k, v in d_ref.items() if v})
    np.testing.assert_array_equal(ref, out)
    r.close()


@pytest.mark.parametrize("name", ["SD15_TINY", "SD35_TINY", "FLUX_TINY"])
def test_reference_real_eval_callback_slicing_on_the_gpu(sd, oracle, gpu, name):
    c = inputs_for(sd, name, np.random.default_rng(14))
    e = sd.Engine(model=c["model"], backend=gpu, flash_attn=True)
    r = rg.RefRunner(e, c["family"], c["version"], gpu, flash_attn=True, overrides=c["overrides"])
    whole = r.compute(c["out"], **c["ref"])
    mm = sd.op_number("MUL_MAT")
    tr = sd.EvalTrace(lambda i, ts: ts.op == mm)
    rg.lib().refg_set_eval_callback(tr._cb, None)
    try:
        sliced = r.compute(c["out"], **c["ref"])
    finally:
        rg.lib().refg_set_eval_callback(sd.EVAL_CALLBACK_FN(), None)
    with sd.EvalTrace(lambda i, ts: ts.op == mm) as tr2:
        host_sliced = c["eng"](e)
    d = rel_l2(sliced, whole)
    print(f"{name}: the reference's own slicing on {gpu}: {len(tr.records)} slices, vs whole graph {d:.2e}; identical to the host restatement's sliced run: "
          f"{np.array_equal(sliced, host_sliced.reshape(sliced.shape))}")
    assert np.isfinite(sliced).all() and d < 5e-3
    np.testing.assert_array_equal(sliced, host_sliced.reshape(sliced.shape))
    assert tr.asked == tr2.asked and len(tr.records) == len(tr2.records) > 30
    for a, b in zip(tr.records, tr2.records):
        assert a[:2] == b[:2]
        np.testing.assert_array_equal(a[3], b[3])
        if a[4] is not None:
            np.testing.assert_array_equal(a[4], b[4])
    r.close()
