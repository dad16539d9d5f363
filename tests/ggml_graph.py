"""Tiny Python front-end over the C ggml API (via ctypes) used by the op-level parity tests:
build a graph once, run it on any registered backend device, fetch named results as numpy."""
from __future__ import annotations

import ctypes as C

import numpy as np

import sdcpp_amd as sd
from sdcpp_amd import F16, F32, BF16, I32, Q4_0, Q8_0, GgmlInitParams, GgmlTensor


def tensor_struct(t) -> GgmlTensor:
    return C.cast(t, C.POINTER(GgmlTensor)).contents


def to_bf16_bits(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)


def encode(arr: np.ndarray, gtype: int) -> bytes:
    """f32 numpy [..., ne0] -> raw bytes of ggml type (I32: integer array as is)."""
    if gtype == I32:
        return np.ascontiguousarray(arr, dtype=np.int32).tobytes()
    a = np.ascontiguousarray(arr, dtype=np.float32)
    if gtype == F32:
        return a.tobytes()
    if gtype == F16:
        return a.astype(np.float16).tobytes()
    if gtype == BF16:
        return to_bf16_bits(a).tobytes()
    L = sd.lib()
    n_per_row = a.shape[-1]
    nrows = a.size // n_per_row
    out = C.create_string_buffer(L.ggml_row_size(gtype, n_per_row) * nrows)
    L.ggml_quantize_chunk(gtype, a.ctypes.data_as(C.c_void_p), out, 0, nrows, n_per_row, None)
    return out.raw


def dequant(arr: np.ndarray, gtype: int) -> np.ndarray:
    """What the stored weights decode to (f32), same shape."""
    a = np.ascontiguousarray(arr, dtype=np.float32)
    if gtype == F32:
        return a
    if gtype == F16:
        return a.astype(np.float16).astype(np.float32)
    raw = encode(a, gtype)
    L = sd.lib()
    n_per_row = a.shape[-1]
    nrows = a.size // n_per_row
    rs = L.ggml_row_size(gtype, n_per_row)
    out = np.empty((nrows, n_per_row), dtype=np.float32)
    buf = C.create_string_buffer(raw, len(raw))
    for r in range(nrows):
        L.ggml_dequantize_row(gtype, C.byref(buf, r * rs), out[r].ctypes.data_as(C.c_void_p), n_per_row)
    return out.reshape(a.shape)


class Graph:
    """with Graph(device) as g:  x = g.input(arr); w = g.weight(arr, F16); y = L.ggml_mul_mat(g.ctx, w, x); out = g.run(y)"""

    def __init__(self, device: str):
        self.L = sd.lib()
        dev = self.L.ggml_backend_dev_by_name(device.encode())
        if not dev:
            raise RuntimeError(f"device {device} not registered (devices: {sd.devices()})")
        self.dev = dev
        self.backend = self.L.ggml_backend_dev_init(dev, None)
        assert self.backend
        self.wctx = self.L.ggml_init(GgmlInitParams(0, None, True))
        self.ctx = self.L.ggml_init(GgmlInitParams(0, None, True))
        self._weights = []   # (tensor, bytes)
        self._inputs = []    # (tensor, bytes)
        self._wbuf = None
        self._galloc = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        L = self.L
        if self._galloc:
            L.ggml_gallocr_free(self._galloc)
            self._galloc = None
        if self._wbuf:
            L.ggml_backend_buffer_free(self._wbuf)
            self._wbuf = None
        if self.ctx:
            L.ggml_free(self.ctx)
            L.ggml_free(self.wctx)
            self.ctx = None
        if self.backend:
            L.ggml_backend_free(self.backend)
            self.backend = None

    def _new(self, ctx, arr: np.ndarray, gtype: int):
        ne = list(reversed(arr.shape))
        assert 1 <= len(ne) <= 4
        fn = getattr(self.L, f"ggml_new_tensor_{len(ne)}d")
        return fn(ctx, gtype, *ne)

    def weight(self, arr: np.ndarray, gtype: int = F16):
        t = self._new(self.wctx, arr, gtype)
        self._weights.append((t, encode(arr, gtype)))
        return t

    def input(self, arr: np.ndarray, gtype: int = F32):
        t = self._new(self.ctx, arr, gtype)
        self.L.ggml_set_input(t)
        self._inputs.append((t, encode(arr, gtype)))
        return t

    def supports(self, node) -> bool:
        return bool(self.L.ggml_backend_dev_supports_op(self.dev, node))

    def run(self, *outs, graph_size: int = 4096):
        L = self.L
        gf = L.ggml_new_graph_custom(self.ctx, graph_size, False)
        for o in outs:
            L.ggml_set_output(o)
            L.ggml_build_forward_expand(gf, o)
        if self._weights and not self._wbuf:
            self._wbuf = L.ggml_backend_alloc_ctx_tensors(self.wctx, self.backend)
            assert self._wbuf
            L.ggml_backend_buffer_set_usage(self._wbuf, 1)  # WEIGHTS
            for t, raw in self._weights:
                L.ggml_backend_tensor_set(t, raw, 0, len(raw))
        self._galloc = L.ggml_gallocr_new(L.ggml_backend_get_default_buffer_type(self.backend))
        assert L.ggml_gallocr_alloc_graph(self._galloc, gf)
        for t, raw in self._inputs:
            if tensor_struct(t).data:  # tensors used only as shape templates (ggml_repeat) are never allocated
                L.ggml_backend_tensor_set(t, raw, 0, len(raw))
        st = L.ggml_backend_graph_compute(self.backend, gf)
        assert st == 0, f"graph_compute status {st}"
        self.n_nodes = L.ggml_graph_n_nodes(gf)
        res = [self.fetch(o) for o in outs]
        return res[0] if len(res) == 1 else res

    def fetch(self, t) -> np.ndarray:
        L = self.L
        ts = tensor_struct(t)
        shape = [int(ts.ne[i]) for i in range(3, -1, -1)]
        nbytes = L.ggml_nbytes(t)
        dt = {F32: np.float32, F16: np.float16}[ts.type]
        out = np.empty(shape, dtype=dt)
        assert out.nbytes == nbytes, (out.nbytes, nbytes, shape)
        L.ggml_backend_tensor_get(t, out.ctypes.data_as(C.c_void_p), 0, nbytes)
        return out.astype(np.float32)
