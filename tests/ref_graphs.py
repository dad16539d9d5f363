"""ctypes front for oracle/_ref/libref_graphs.so — the REFERENCE's own graph builders and runner (src/core/ggml_extend.hpp GGMLRunner + op wrappers,
src/model/diffusion/{unet,mmdit,flux}.hpp, src/model/vae/auto_encoder_kl.hpp, ...) compiled from /root/reference against this repository's ggml front-end
(oracle/Makefile, oracle/ref_graphs_wrap.cpp).  TEST infrastructure: used by tests/test_ref_graphs.py (CPU) and tests/test_gpu_ref_graphs.py.

The library is built where /root/reference exists (this container; `__graft_entry__.build()`), is git-ignored and travels to the GPU box with the snapshot."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import sdcpp_amd as sd

LIB = Path(__file__).resolve().parent.parent / "oracle" / "_ref" / "libref_graphs.so"
FAMILY = {"unet": 0, "vae": 1, "mmdit": 2, "flux": 3, "tae": 4, "vae_enc": 5}
_lib = None


def available() -> bool:
    return LIB.exists()


def lib():
    global _lib
    if _lib is None:
        sd.lib()  # libsdcpp-host.so first, RTLD_GLOBAL: the reference code resolves its ggml symbols there
        L = C.CDLL(str(LIB))
        L.refg_storage_add.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.refg_new.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]
        L.refg_new.restype = C.c_void_p
        L.refg_free.argtypes = [C.c_void_p]
        L.refg_param_count.argtypes = [C.c_void_p]
        L.refg_param_count.restype = C.c_int64
        L.refg_param_name.argtypes = [C.c_void_p, C.c_int64]
        L.refg_param_name.restype = C.c_char_p
        L.refg_param_tensor.argtypes = [C.c_void_p, C.c_int64]
        L.refg_param_tensor.restype = C.c_void_p
        L.refg_alloc_params.argtypes = [C.c_void_p]
        L.refg_set_conv2d_scale.argtypes = [C.c_void_p, C.c_float]
        L.refg_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                               C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.refg_run.restype = C.c_int64
        L.refg_description.argtypes = [C.c_void_p]
        L.refg_description.restype = C.c_char_p
        L.refg_set_eval_callback.argtypes = [sd.EVAL_CALLBACK_FN, C.c_void_p]
        L.refg_set_eval_callback.restype = None
        _lib = L
    return _lib


def _n_dims(name: str, ne) -> int:
    """Dimension count a checkpoint file would record for this tensor (the engine's table keeps ggml's 4 extents): trailing 1s dropped, except the
    MMDiT position table, which the reference's config detection expects as 3-D [hidden, patches, 1] (mmdit.hpp:66)."""
    if name.endswith("pos_embed"):
        return 3
    n = 4
    while n > 1 and ne[n - 1] == 1:
        n -= 1
    return n


# configuration fields of the reduced-width test models that no weight shape reveals (csrc/host/models.hpp: UNetConfig::tiny, MMDiTConfig::tiny, FluxConfig::tiny)
OVERRIDES = {
    "SD15_TINY": "num_heads=2;context_dim=64",
    "SDXL_TINY": "num_head_channels=16;transformer_depth=1,1,2;context_dim=64",
    "SD35_TINY": "depth=3",
    "SD3M_TINY": "depth=3",
    "FLUX_TINY": "vec_in_dim=64;axes_dim=8,12,12",
}


class RefRunner:
    """One of the reference's runners (UNetModelRunner / AutoEncoderKL / MMDiTRunner / Flux::FluxRunner) over the SAME weights as `engine`: the storage
    table (names, types, shapes) is the engine's tensor table, the values are copied tensor by tensor (raw bytes) into the reference's parameters."""

    def __init__(self, engine, family: str, version: str, device: str, flash_attn: bool = False, overrides: str = "", copy_weights: bool = True):
        L, H = lib(), sd.lib()
        self.L, self.H, self.engine, self.family = L, H, engine, family
        prefix = "first_stage_model" if family in ("vae", "vae_enc") else "model.diffusion_model"
        L.refg_storage_clear()
        for name in engine.tensor_names():
            ne, gtype, _ = engine.tensor_info(name)
            L.refg_storage_add(name.encode(), gtype, _n_dims(name, ne), (C.c_int64 * 4)(*ne))
        dev = H.ggml_backend_dev_by_name(device.encode())
        assert dev, f"device {device} not registered"
        self.backend = H.ggml_backend_dev_init(dev, None)
        assert self.backend
        self.h = L.refg_new(FAMILY[family], version.encode(), self.backend, prefix.encode(), int(flash_attn), overrides.encode())
        assert self.h
        self.param_names = [L.refg_param_name(self.h, i).decode() for i in range(L.refg_param_count(self.h))]
        if copy_weights:
            assert L.refg_alloc_params(self.h)
            H.sd_get_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
            H.sd_get_tensor.restype = C.c_bool
            for i, name in enumerate(self.param_names):
                t = L.refg_param_tensor(self.h, i)
                nbytes = H.ggml_nbytes(t)
                ne, gtype, eng_bytes = engine.tensor_info(name)
                assert eng_bytes == nbytes, (name, eng_bytes, nbytes)
                buf = C.create_string_buffer(nbytes)
                assert H.sd_get_tensor(engine._ctx, name.encode(), buf, nbytes), name
                H.ggml_backend_tensor_set(t, buf, 0, nbytes)

    def close(self):
        if self.h:
            self.L.refg_free(self.h)
            self.h = None
        if self.backend:
            self.H.ggml_backend_free(self.backend)
            self.backend = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_conv2d_scale(self, s: float):
        self.L.refg_set_conv2d_scale(self.h, s)

    def _run(self, describe: bool, x, t=None, ctx=None, y=None, guidance=None, out_shape=None):
        def arr(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.float32)

        def ne_of(a, n):  # numpy [.., ne1, ne0] -> ggml extents
            sh = list(reversed(a.shape))
            assert len(sh) <= n
            return (C.c_int64 * n)(*(sh + [1] * (n - len(sh))))

        x, t, ctx, y, guidance = arr(x), arr(t), arr(ctx), arr(y), arr(guidance)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        out = np.empty(out_shape, dtype=np.float32) if (out_shape is not None and not describe) else None
        n = self.L.refg_run(self.h, int(describe), ptr(x), ne_of(x, 4), ptr(t), 0 if t is None else t.size, ptr(ctx), None if ctx is None else ne_of(ctx, 3), ptr(y),
                            None if y is None else ne_of(y, 2), ptr(guidance), 0 if guidance is None else guidance.size, ptr(out), 0 if out is None else out.size)
        assert n >= 0, "the reference runner failed (see its log above)"
        if describe:
            return self.L.refg_description(self.h).decode()
        assert n == out.size, (n, out.shape)
        return out

    def describe(self, x, t=None, ctx=None, y=None, guidance=None) -> str:
        return self._run(True, x, t, ctx, y, guidance)

    def compute(self, out_shape, x, t=None, ctx=None, y=None, guidance=None) -> np.ndarray:
        return self._run(False, x, t, ctx, y, guidance, out_shape)


def engine_graph_description(run, compute: bool = True) -> str:
    """Description (sdm_graph_describe) of the LAST graph the engine computes inside run(); compute=False: the graph is built, placed and described, not run."""
    H = sd.lib()
    H.sdm_set_graph_capture.argtypes = [C.c_int]
    H.sdm_last_graph_description.argtypes = [C.c_char_p, C.c_size_t]
    H.sdm_last_graph_description.restype = C.c_size_t
    H.sdm_set_graph_capture(1 if compute else 2)
    try:
        try:
            run()
        except sd.EngineError as e:
            if compute or "graph captured" not in str(e):
                raise
        n = H.sdm_last_graph_description(None, 0)
        buf = C.create_string_buffer(n)
        H.sdm_last_graph_description(buf, n)
    finally:
        H.sdm_set_graph_capture(0)
    return buf.value.decode()


def split_description(desc: str):
    leafs, nodes = [], []
    for line in desc.splitlines():
        (leafs if line.startswith("L ") else nodes).append(line)
    return leafs, nodes


def strip_name(line: str) -> str:
    return line[: line.index(" name=")]
