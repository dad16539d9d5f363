"""Checkpoint files -> MI355X buffers: a safetensors file (F32 / F16 / BF16 tensors) and GGUF files with
q8_0 / q4_0 Linear weights (the raw block bytes of src/model_io/gguf_io.cpp's types) are written here from the formats' public specifications,
loaded through sd_load_weights into engines on the MI355X backend AND on the CPU oracle, and checked
  * tensor by tensor: what the GPU buffers hold (read back through the backend) equals what the file encodes, converted per the load-time rules
    (model_loader.cpp:155-205: file dtype -> f32 -> parameter type; quantised blocks byte for byte);
  * through a UNet forward: the kernels consume the loaded bytes (raw q8_0 / q4_0 blocks included) and agree with the oracle run on the same file.
test_model_io.py covers the readers' edge cases on the CPU; this file is the device half."""
import numpy as np
import pytest

from ggml_graph import Q4_0, Q8_0, dequant, encode
from test_model_io import _f32_to_bf16_bits, _names, _write_gguf, _write_safetensors

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _shape(e, n):
    ne, ty, _ = e.tensor_info(n)
    shape = tuple(int(d) for d in reversed(ne))
    while len(shape) > 1 and shape[0] == 1:
        shape = shape[1:]
    return shape, ty


def test_safetensors_file_into_gpu_buffers(sd, oracle, gpu, tmp_path):
    e_gpu = sd.Engine(model=sd.SD15_TINY, backend=gpu)
    e_ref = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    names = _names(e_gpu)
    rng = np.random.default_rng(31)
    tensors, want = {}, {}
    for i, n in enumerate(names):
        shape, _ = _shape(e_gpu, n)
        fan = max(1, int(np.prod(shape[1:])) if len(shape) > 1 else 1)
        a = (rng.standard_normal(shape) / np.sqrt(fan)).astype(np.float32) if len(shape) > 1 else (rng.standard_normal(shape) * 0.1 + (1.0 if n.endswith("norm.weight") else 0.0)).astype(np.float32)
        kind = i % 3
        if kind == 0:
            tensors[n], want[n] = ("F32", a), a
        elif kind == 1:
            tensors[n], want[n] = ("F16", a.astype(np.float16)), a.astype(np.float16).astype(np.float32)
        else:
            bits = _f32_to_bf16_bits(a)
            tensors[n], want[n] = ("BF16", bits), (bits.astype(np.uint32) << 16).view(np.float32)
    p = tmp_path / "tiny_gpu.safetensors"
    _write_safetensors(p, tensors, {"format": "pt"})
    for e in (e_gpu, e_ref):
        assert e.load_weights(p) == {"loaded": len(names), "missing": 0, "unused": 0}
    for n in names[::7]:   # device -> host read-back of every 7th tensor
        _, ty = _shape(e_gpu, n)
        ref = want[n].ravel()
        if ty == sd.F16:
            ref = ref.astype(np.float16).astype(np.float32)
        np.testing.assert_array_equal(e_gpu.get_tensor(n).ravel(), ref, err_msg=n)
    x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
    ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
    t = np.array([321.0, 321.0], np.float32)
    out, ref = e_gpu.unet_forward(x, t, ctx), e_ref.unet_forward(x, t, ctx)
    err = rel_l2(out, ref)
    print(f"UNet forward on weights loaded from a safetensors file: GPU vs oracle rel-L2 {err:.3e}")
    assert np.isfinite(out).all() and err < 5e-3


@pytest.mark.parametrize("qname,qtype,tol", [("Q8_0", Q8_0, 2e-2), ("Q4_0", Q4_0, 8e-2)])
def test_gguf_quantised_blocks_into_gpu_buffers(sd, oracle, gpu, tmp_path, qname, qtype, tol):
    """Every quantisable Linear weight arrives as raw q8_0 / q4_0 blocks from the file; the GPU kernels read those bytes (k_qgemv / k_qgemm16 on few
    rows, the dequantised f16 image above).  The oracle quantises activations as ggml-cpu does, hence the looser forward bar; against the same
    engine with dequantised f32 weights... is covered at full width in test_zz_gpu_fullsize.py."""
    wt = getattr(sd, qname)
    e_gpu = sd.Engine(model=sd.SD15_TINY, backend=gpu, wtype=wt)
    e_ref = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=wt)
    rng = np.random.default_rng(32)
    tensors, want, nq = [], {}, 0
    for n in _names(e_gpu):
        ne, pty, _ = e_gpu.tensor_info(n)
        ne = [int(d) for d in ne]
        while len(ne) > 1 and ne[-1] == 1:
            ne = ne[:-1]
        shape = tuple(reversed(ne))
        a = (rng.standard_normal(shape) / np.sqrt(max(1, int(np.prod(ne[:-1]))))).astype(np.float32) if len(ne) > 1 else (rng.standard_normal(shape) * 0.1 + (1.0 if n.endswith("norm.weight") else 0.0)).astype(np.float32)
        if pty == qtype:   # quantised parameter: the file carries the blocks, the buffer must hold them byte for byte
            raw = encode(a, qtype)
            tensors.append((n, qtype, ne, raw))
            want[n] = dequant(a, qtype)
            nq += 1
        elif pty == sd.F16:
            tensors.append((n, sd.F16, ne, a.astype(np.float16).tobytes()))
            want[n] = a.astype(np.float16).astype(np.float32)
        else:
            tensors.append((n, sd.F32, ne, a.tobytes()))
            want[n] = a
    assert nq > 20
    p = tmp_path / f"tiny_{qname}.gguf"
    _write_gguf(p, tensors)
    for e in (e_gpu, e_ref):
        r = e.load_weights(p)
        assert r["loaded"] == len(tensors) and r["missing"] == 0
    for n, _, _, _ in tensors[::5]:
        np.testing.assert_allclose(e_gpu.get_tensor(n).ravel(), want[n].ravel(), rtol=0, atol=1e-7, err_msg=n)
    x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
    ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
    t = np.array([321.0, 321.0], np.float32)
    out, ref = e_gpu.unet_forward(x, t, ctx), e_ref.unet_forward(x, t, ctx)
    err = rel_l2(out, ref)
    print(f"UNet forward on {qname} blocks loaded from a GGUF file: GPU vs oracle rel-L2 {err:.3e}")
    assert np.isfinite(out).all() and err < tol


def test_torch_checkpoint_and_k_quant_gguf_into_gpu_buffers(sd, oracle, gpu, tmp_path):
    """A whole tiny model as a torch.save checkpoint (Lightning-style wrapper, f16 / f32 / bf16 tensors) into MI355X buffers, then two of its Linear
    weights overwritten from a GGUF file holding q4_K / q6_K super-blocks (decoded at load, model_io.hpp): read back and run through a UNet forward
    against the oracle engine loaded from the same two files."""
    import collections
    import torch
    from test_model_io import _kquant_random_blocks
    e_gpu = sd.Engine(model=sd.SD15_TINY, backend=gpu)
    e_ref = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    names = _names(e_gpu)
    rng = np.random.default_rng(33)
    sdict, want = collections.OrderedDict(), {}
    for i, n in enumerate(names):
        shape, _ = _shape(e_gpu, n)
        fan = max(1, int(np.prod(shape[1:])) if len(shape) > 1 else 1)
        a = (rng.standard_normal(shape) / np.sqrt(fan)).astype(np.float32) if len(shape) > 1 else (rng.standard_normal(shape) * 0.1 + (1.0 if n.endswith("norm.weight") else 0.0)).astype(np.float32)
        t = torch.from_numpy(a).to((torch.float32, torch.float16, torch.bfloat16)[i % 3])
        sdict[n], want[n] = t, t.to(torch.float32).numpy()
    p = tmp_path / "tiny.ckpt"
    torch.save({"global_step": 1, "state_dict": sdict, "optimizer_states": [{"state": {0: {"exp_avg": torch.zeros(4, 4)}}}]}, p)
    kq = [("model.diffusion_model.input_blocks.4.1.transformer_blocks.0.ff.net.2.weight", "q4_K", 12), ("model.diffusion_model.input_blocks.5.1.transformer_blocks.0.ff.net.2.weight", "q6_K", 14)]
    gg = []
    for n, kind, gtype in kq:
        ne = [int(d) for d in e_gpu.tensor_info(n)[0][:2]]
        raw, val = _kquant_random_blocks(rng, kind, ne[0] * ne[1] // 256)
        gg.append((n, gtype, ne, raw))
        want[n] = val.reshape(ne[1], ne[0])
    _write_gguf(tmp_path / "kq.gguf", gg)
    for e in (e_gpu, e_ref):
        assert e.load_weights(p) == {"loaded": len(names), "missing": 0, "unused": 0}
        assert e.load_weights(tmp_path / "kq.gguf")["loaded"] == 2
    for n in names[::9] + [k[0] for k in kq]:
        _, ty = _shape(e_gpu, n)
        ref = want[n].ravel()
        if ty == sd.F16:
            np.testing.assert_allclose(e_gpu.get_tensor(n).ravel(), ref.astype(np.float16).astype(np.float32), rtol=1e-3, atol=1e-7, err_msg=n)
        else:
            np.testing.assert_allclose(e_gpu.get_tensor(n).ravel(), ref, rtol=2e-7, atol=1e-7, err_msg=n)
        np.testing.assert_array_equal(e_gpu.get_tensor(n), e_ref.get_tensor(n), err_msg=n)  # both engines hold the same converted bytes
    x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
    ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
    t = np.array([500.0, 500.0], np.float32)
    out, ref = e_gpu.unet_forward(x, t, ctx), e_ref.unet_forward(x, t, ctx)
    err = rel_l2(out, ref)
    print(f"UNet forward on weights from a torch checkpoint + K-quant GGUF: GPU vs oracle rel-L2 {err:.3e}")
    assert np.isfinite(out).all() and err < 5e-3
