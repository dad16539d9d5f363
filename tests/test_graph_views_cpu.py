"""Host side of the sub-graph-view contract, on the CPU (the GPU half: tests/test_gpu_graph_views.py).

* the graph front-end fills ggml_cgraph::visited_hash_set / use_counts the way upstream ggml does (hash = address >> 4, linear probing, `used` bitset; one
  use per source slot per visited tensor) — what a backend handed a view (sd_ggml_graph_view, src/core/ggml_extend_backend.cpp:449-463) reads;
* sdm_backend_graph_compute_with_eval_callback restates sd_backend_graph_compute_with_eval_callback (:466-509): every node asked once, in order; the
  graph cut behind each wanted node; ABORTED when the callback returns false; the sliced result equals the whole-graph result bit for bit on a backend that
  executes node by node (the oracle).
"""
import ctypes as C

import numpy as np
import pytest

from ggml_graph import Graph, tensor_struct


class HashSet(C.Structure):
    _fields_ = [("size", C.c_size_t), ("used", C.POINTER(C.c_uint32)), ("keys", C.POINTER(C.c_void_p))]


class CGraph(C.Structure):
    """struct ggml_cgraph (include/ggml-abi.h)"""
    _fields_ = [("size", C.c_int), ("n_nodes", C.c_int), ("n_leafs", C.c_int), ("nodes", C.POINTER(C.c_void_p)), ("grads", C.c_void_p),
                ("grad_accs", C.c_void_p), ("leafs", C.POINTER(C.c_void_p)), ("use_counts", C.POINTER(C.c_int32)), ("visited_hash_set", HashSet),
                ("order", C.c_int), ("uid", C.c_uint64)]


def test_use_counts_and_hash_set_follow_the_upstream_layout(sd, oracle):
    L = sd.lib()
    rng = np.random.default_rng(0)
    with Graph(oracle) as g:
        x = g.input(rng.standard_normal((5, 16)).astype(np.float32))
        w = g.weight(rng.standard_normal((24, 16)).astype(np.float32), sd.F16)
        b = g.weight(rng.standard_normal((24,)).astype(np.float32), sd.F32)
        h = L.ggml_add_inplace(g.ctx, L.ggml_mul_mat(g.ctx, w, x), b)
        lo = L.ggml_view_2d(g.ctx, h, 12, 5, 24 * 4, 0)
        hi = L.ggml_view_2d(g.ctx, h, 12, 5, 24 * 4, 12 * 4)
        y = L.ggml_mul(g.ctx, lo, L.ggml_gelu(g.ctx, L.ggml_cont(g.ctx, hi)))
        z = L.ggml_add(g.ctx, y, y)  # one tensor in two source slots of one node: two uses
        gf = L.ggml_new_graph_custom(g.ctx, 64, False)
        L.ggml_build_forward_expand(gf, z)
        cg = C.cast(gf, C.POINTER(CGraph)).contents
        assert cg.size == 64 and cg.n_nodes == 8 and cg.uid != 0 and bool(cg.leafs) and bool(cg.use_counts)
        hs = cg.visited_hash_set
        assert hs.size >= 128

        def slot(ptr):
            i = h0 = (ptr >> 4) % hs.size
            while (hs.used[i >> 5] >> (i & 31)) & 1:
                if hs.keys[i] == ptr:
                    return i
                i = (i + 1) % hs.size
                assert i != h0
            return None

        nodes = [cg.nodes[i] for i in range(cg.n_nodes)]
        leafs = [cg.leafs[i] for i in range(cg.n_leafs)]
        assert len(leafs) == 3
        expect = {p: 0 for p in nodes + leafs}
        for p in nodes:
            for s in tensor_struct(p).src:
                if s:
                    expect[s] += 1
        for p, n in expect.items():
            i = slot(p)
            assert i is not None, "every visited tensor is a key of the set"
            assert cg.use_counts[i] == n
        assert expect[y] == 2 and expect[h] == 2 and expect[z] == 0
        n_used = sum(bin(hs.used[k]).count("1") for k in range((hs.size + 31) // 32))
        assert n_used == len(expect)


@pytest.mark.parametrize("name", ["SD15_TINY", "FLUX_TINY"])
def test_sliced_evaluation_equals_whole_graph_on_a_node_by_node_backend(sd, oracle, name):
    rng = np.random.default_rng(3)
    if name == "SD15_TINY":
        x = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
        args = (x, np.array([500.0], np.float32), rng.standard_normal((1, 77, 64)).astype(np.float32), None)
    else:
        x = rng.standard_normal((1, 16, 8, 6)).astype(np.float32)
        args = (x, np.array([0.5], np.float32), rng.standard_normal((1, 12, 96)).astype(np.float32), rng.standard_normal((1, 64)).astype(np.float32))
    e = sd.Engine(model=getattr(sd, name), backend=oracle, flash_attn=True)
    whole = e.unet_forward(*args)
    mm = sd.op_number("MUL_MAT")
    with sd.EvalTrace(lambda i, ts: ts.op == mm) as tr:
        out = e.unet_forward(*args)
    np.testing.assert_array_equal(out, whole)
    assert tr.graphs == 1 and tr.asked == e.stats()["graph_nodes"]
    assert [r[0] for r in tr.records] == sorted(r[0] for r in tr.records) and all(r[1] == mm for r in tr.records)
    assert all(r[3] is not None and np.isfinite(r[3]).all() for r in tr.records)
    assert sum(r[4] is not None for r in tr.records) == len(tr.records)  # src[1] of every MUL_MAT was readable (imatrix)
    # without a callback installed the graph goes through plain graph_compute again
    np.testing.assert_array_equal(e.unet_forward(*args), whole)


def test_callback_returning_false_gives_status_aborted(sd, oracle):
    rng = np.random.default_rng(4)
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    args = (rng.standard_normal((1, 4, 8, 8)).astype(np.float32), np.array([500.0], np.float32), rng.standard_normal((1, 77, 64)).astype(np.float32), None)
    with sd.EvalTrace(lambda i, ts: i % 50 == 49, stop_after=2) as tr:
        with pytest.raises(sd.EngineError, match="(?i)abort"):
            e.unet_forward(*args)
    assert len(tr.records) == 2 and tr.asked == 100  # nothing behind the stopping node was asked about or computed
    assert np.isfinite(e.unet_forward(*args)).all()
