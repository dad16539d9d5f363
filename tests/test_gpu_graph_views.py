"""Sub-graph views at the drop-in boundary.

The reference evaluates a graph node by node when a callback is installed (sd_set_backend_eval_callback, include/stable-diffusion.h:442-447; the imatrix
collector, src/runtime/imatrix.cpp:39-100): sd_backend_graph_compute_with_eval_callback (src/core/ggml_extend_backend.cpp:466-509) cuts the graph BEHIND
every node the callback asks for and hands graph_compute each slice as a view (sd_ggml_graph_view, :449-463: nodes + i0, leafs NULL, size 0, uid 0, the
PARENT's use_counts / visited_hash_set).  A slice is not a closed graph: a tensor with no reader inside it may be read by a later slice or by the callback,
so a fusing backend must materialise it.  The host side here (sdm_backend_graph_compute_with_eval_callback, csrc/host/engine.cpp) restates that loop and
view constructor statement for statement; every test below goes through it and through the plug-in's vtable.

For the tiny UNet (SD1.5 and SDXL topologies), KL-VAE decoder, MMDiT and FLUX graphs, with the graph cut
  (a) behind EVERY MUL_MAT (the imatrix pattern: the callback reads the node and src[1], the activations), and
  (b) behind 20 random nodes (three seeds),
the sliced run on the MI355X backend must
  * give the whole-graph result within the parity bar of the whole-graph tests (5e-3 rel-L2; measured 1e-3 with ~300 cuts, less with 20).  It cannot be
    bit-identical: a cut INSIDE a chain the whole graph runs as one kernel (IM2COL | MUL_MAT | CONT of a conv, q·k | softmax | ·v of an attention) makes
    the slices run the chain's nodes one by one — the im2col matrix in f16 times the kernel on the exact-f32 MFMA path instead of the implicit-GEMM conv,
    f32 scores instead of the flash kernel's f16 probabilities — i.e. other, equally valid roundings of the same reference arithmetic;
  * hand the callback, for every node it asked for (and for src[1] of every MUL_MAT), what the CPU oracle — which executes node by node, so a view is just
    a shorter node list to it — computes for that node under the same slicing: the parity bar of the whole-graph tests, per tensor.
Before round 6 the planner assumed every graph closed: slice 1 of (a) wrote only the f16 operand image of a LayerNorm whose other readers (the k / v
projections) sat in slice 2, which then read an f32 tensor nobody had written.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _case(sd, name):
    rng = np.random.default_rng(sum(name.encode()))
    if name in ("SD15_TINY", "SDXL_TINY"):
        x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
        t = np.array([731.0, 210.0], dtype=np.float32)
        ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
        y = rng.standard_normal((1, 96)).astype(np.float32) if "XL" in name else None
        return getattr(sd, name), lambda e: e.unet_forward(x, t, ctx, y)
    if name == "VAE":
        z = rng.standard_normal((1, 4, 12, 10)).astype(np.float32) * 0.5
        return sd.SD15_TINY, lambda e: e.vae_decode(z)
    if name == "VAE_SCALED":  # SDXL: Conv2d scale 1/32, SCALE nodes folded into the conv (both must survive a cut between SCALE and IM2COL)
        z = rng.standard_normal((1, 4, 12, 10)).astype(np.float32) * 0.5
        return sd.SDXL_TINY, lambda e: e.vae_decode(z)
    if name == "TAE":  # TAESD's conv -> ReLU -> conv chains: the ReLUs folded into the operand-image pack must survive a cut between the ReLU and its conv
        z = rng.standard_normal((2, 4, 12, 10)).astype(np.float32) * 1.5
        return sd.SD15_TINY, lambda e: e.tae_decode(z)
    if name == "VAE_ENC":  # the encode graph: PAD + stride-2 downsample convs
        img = rng.random((1, 3, 48, 40)).astype(np.float32)
        return sd.SD15_TINY, lambda e: e.vae_encode(img, return_moments=True)[1]
    x = rng.standard_normal((2, 16, 14, 12)).astype(np.float32)
    ctx = rng.standard_normal((1, 40, 96)).astype(np.float32)
    y = rng.standard_normal((1, 64)).astype(np.float32)
    if name == "SD35_TINY":
        t = np.array([731.0, 210.0], dtype=np.float32)
    else:
        t = np.array([0.81, 0.27], dtype=np.float32)
    return getattr(sd, name), lambda e: e.unet_forward(x, t, ctx, y)


CASES = ["SD15_TINY", "SDXL_TINY", "VAE", "VAE_SCALED", "SD35_TINY", "FLUX_TINY", "TAE", "VAE_ENC"]


def _traced(sd, engine, run, want):
    with sd.EvalTrace(want) as tr:
        out = run(engine)
    return out, tr


def _compare_records(name, what, got, ref, tol):
    assert len(got.records) == len(ref.records) and got.asked == ref.asked, (len(got.records), len(ref.records), got.asked, ref.asked)
    worst = (0.0, None)
    checked = 0
    for (gi, gop, gname, gv, gs1), (ri, rop, rname, rv, rs1) in zip(got.records, ref.records):
        assert (gi, gop) == (ri, rop)
        for tag, a, b in (("node", gv, rv), ("src1", gs1, rs1)):
            if a is None or b is None:
                assert (a is None) == (b is None)
                continue
            assert a.shape == b.shape
            assert np.isfinite(a).all(), f"{name} {what}: node {gi} ({gname}) {tag} not finite on the GPU"
            scale = float(np.sqrt(np.mean(b.astype(np.float64) ** 2)))
            if scale < 1e-12:
                continue
            e = rel_l2(a, b)
            checked += 1
            if e > worst[0]:
                worst = (e, f"node {gi} op {gop} {gname} {tag}")
            assert e < tol, f"{name} {what}: node {gi} op {gop} ({gname}) {tag}: rel-L2 {e:.3e} vs the oracle under the same slicing"
    print(f"{name} {what}: {len(got.records)} callback nodes, {checked} tensors compared, worst {worst[0]:.2e} ({worst[1]})")
    return checked


@pytest.mark.parametrize("name", CASES)
def test_views_cut_behind_every_mul_mat(sd, oracle, gpu, name):
    model, run = _case(sd, name)
    mm = sd.op_number("MUL_MAT")
    gpu_e = sd.Engine(model=model, backend=gpu, flash_attn=True)
    ref_e = sd.Engine(model=model, backend=oracle, flash_attn=True)
    whole = run(gpu_e)
    want = lambda i, ts: ts.op == mm
    st0 = sd.backend_stats() if gpu != oracle else None
    out, tr = _traced(sd, gpu_e, run, want)
    st1 = sd.backend_stats() if gpu != oracle else None
    ref_out, ref_tr = _traced(sd, ref_e, run, want)
    assert tr.records and tr.graphs == 1
    if gpu != oracle:
        assert st1["view_graphs"] - st0["view_graphs"] == len(tr.records) + (0 if tr.records[-1][0] == tr.asked - 1 else 1)  # one plan per slice
        assert st1["view_external_nodes"] > st0["view_external_nodes"]
    assert np.isfinite(out).all()
    d = rel_l2(out, whole)
    print(f"{name}: sliced behind {len(tr.records)} MUL_MATs of {tr.asked} nodes: sliced vs whole graph rel-L2 {d:.2e} (bit-identical: {np.array_equal(out, whole)})")
    assert d < 5e-3
    assert rel_l2(out, ref_out) < 5e-3
    tol = 2e-2 if name == "FLUX_TINY" else 1e-2
    assert _compare_records(name, "every MUL_MAT", tr, ref_tr, tol) >= len(tr.records)
    # the whole graph again, no callback: the slices' plans must not have disturbed the cached whole-graph plan
    np.testing.assert_array_equal(run(gpu_e), whole)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_views_cut_at_random_nodes(sd, oracle, gpu, name, seed):
    model, run = _case(sd, name)
    gpu_e = sd.Engine(model=model, backend=gpu, flash_attn=(seed != 2))
    ref_e = sd.Engine(model=model, backend=oracle, flash_attn=(seed != 2))
    whole = run(gpu_e)
    _, count = _traced(sd, ref_e, run, lambda i, ts: False)  # a callback that wants nothing: one slice = the whole graph; counts the nodes
    n_nodes = count.asked
    assert not count.records and n_nodes > 40
    cuts = set(np.random.default_rng(1000 * seed + len(name)).choice(n_nodes - 1, size=20, replace=False).tolist())
    want = lambda i, ts: i in cuts
    out, tr = _traced(sd, gpu_e, run, want)
    ref_out, ref_tr = _traced(sd, ref_e, run, want)
    assert len(tr.records) == 20 and tr.asked == n_nodes
    d = rel_l2(out, whole)
    print(f"{name} seed {seed}: 20 random cuts in {n_nodes} nodes: sliced vs whole rel-L2 {d:.2e} (bit-identical: {np.array_equal(out, whole)})")
    assert np.isfinite(out).all() and d < 5e-3
    assert rel_l2(out, ref_out) < 5e-3
    _compare_records(name, f"random cuts seed {seed}", tr, ref_tr, 2e-2 if name == "FLUX_TINY" else 1e-2)


def test_view_without_use_counts_is_treated_as_fully_external(sd, oracle, gpu):
    """A host whose ggml has no use_counts table (older ggml: the field is NULL in the view): every node of a slice counts as needed outside it.
    GGML_MI355X_IGNORE_USE_COUNTS=1 makes the plug-in behave as if the table were absent."""
    if gpu == oracle:
        pytest.skip("plug-in option")
    model, run = _case(sd, "SD15_TINY")
    mm = sd.op_number("MUL_MAT")
    e = sd.Engine(model=model, backend=gpu, flash_attn=True)
    whole = run(e)
    sd.backend_set_option("ignore_use_counts", 1)
    try:
        out, tr = _traced(sd, e, run, lambda i, ts: ts.op == mm)
    finally:
        sd.backend_set_option("ignore_use_counts", 0)
    assert rel_l2(out, whole) < 5e-3


def test_callback_returning_false_aborts_the_graph(sd, oracle, gpu):
    model, run = _case(sd, "SD15_TINY")
    e = sd.Engine(model=model, backend=gpu, flash_attn=True)
    mm = sd.op_number("MUL_MAT")
    with sd.EvalTrace(lambda i, ts: ts.op == mm, stop_after=3) as tr:
        with pytest.raises(sd.EngineError, match="(?i)abort"):
            run(e)
    assert len(tr.records) == 3
    assert np.isfinite(run(e)).all()  # the engine is usable afterwards
