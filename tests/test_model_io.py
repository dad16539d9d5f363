"""CPU suite: checkpoint readers — safetensors and GGUF files written here from the formats' public
specifications, loaded through sd_load_weights into the tiny SD1.5 engine, and checked tensor by tensor and through a UNet forward."""
import json
import struct

import numpy as np
import pytest


def _names(e):
    L = __import__("sdcpp_amd").lib()
    return [L.sd_tensor_name(e._ctx, i).decode() for i in range(L.sd_tensor_count(e._ctx))]


def _write_safetensors(path, tensors, metadata=None):
    """tensors: {name: (dtype_str, np array in torch order)}; BF16 arrays are passed as uint16 bit patterns."""
    header, blobs, off = {}, [], 0
    if metadata:
        header["__metadata__"] = metadata
    for name, (dt, arr) in tensors.items():
        raw = np.ascontiguousarray(arr).tobytes()
        header[name] = {"dtype": dt, "shape": list(arr.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)


def _f32_to_bf16_bits(a):
    u = a.astype(np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _q8_0_blocks(a):
    """ggml block_q8_0: f16 d + 32 int8, d = amax / 127 (Appendix D of SURVEY.md)"""
    a = a.astype(np.float32).reshape(-1, 32)
    d = np.abs(a).max(axis=1) / 127.0
    idd = np.where(d > 0, 1.0 / np.where(d > 0, d, 1), 0.0)
    q = np.round(a * idd[:, None]).astype(np.int8)
    out = bytearray()
    for i in range(a.shape[0]):
        out += np.float16(d[i]).tobytes() + q[i].tobytes()
    return bytes(out), (q.astype(np.float32) * np.float16(d).astype(np.float32)[:, None]).reshape(-1)


def _write_gguf(path, tensors, alignment=32):
    """tensors: [(name, ggml_type, ne (ggml order), raw bytes)] — GGUF v3"""
    def s(x):
        b = x.encode()
        return struct.pack("<Q", len(b)) + b

    kv = s("general.architecture") + struct.pack("<I", 8) + s("sd") + s("general.alignment") + struct.pack("<II", 4, alignment)
    infos, data, off = b"", b"", 0
    for name, ty, ne, raw in tensors:
        infos += s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", int(d)) for d in ne) + struct.pack("<IQ", ty, off)
        pad = (alignment - len(raw) % alignment) % alignment
        data += raw + b"\0" * pad
        off += len(raw) + pad
    head = b"GGUF" + struct.pack("<IQQ", 3, len(tensors), 2) + kv + infos
    head += b"\0" * ((alignment - len(head) % alignment) % alignment)
    with open(path, "wb") as f:
        f.write(head + data)


def test_safetensors_load_converts_and_runs(sd, oracle, tmp_path):
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    names = _names(e)
    rng = np.random.default_rng(3)
    tensors, want = {}, {}
    for i, n in enumerate(names):
        ne, ty, _ = e.tensor_info(n)
        shape = tuple(int(d) for d in reversed(ne))
        while len(shape) > 1 and shape[0] == 1:
            shape = shape[1:]
        a = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        kind = i % 3
        if kind == 0:
            tensors[n] = ("F32", a)
            want[n] = a
        elif kind == 1:
            tensors[n] = ("F16", a.astype(np.float16))
            want[n] = a.astype(np.float16).astype(np.float32)
        else:
            bits = _f32_to_bf16_bits(a)
            tensors[n] = ("BF16", bits)
            want[n] = (bits.astype(np.uint32) << 16).view(np.float32)
    tensors["some.unrelated.tensor"] = ("F32", np.zeros((3,), np.float32))
    p = tmp_path / "tiny.safetensors"
    _write_safetensors(p, tensors, {"format": "pt"})
    r = e.load_weights(p)
    assert r == {"loaded": len(names), "missing": 0, "unused": 1}
    for n in names[:40] + names[-40:]:
        _, ty, _ = e.tensor_info(n)
        got = e.get_tensor(n).ravel()
        ref = want[n].ravel()
        if ty == sd.F16:
            ref = ref.astype(np.float16).astype(np.float32)   # file dtype -> f32 -> parameter type (model_loader.cpp:155-205)
        np.testing.assert_array_equal(got, ref, err_msg=n)
    # the loaded model runs and depends on the loaded values
    x = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
    y1 = e.unet_forward(x, np.array([100.0], np.float32), ctx)
    assert np.isfinite(y1).all()
    e2 = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    assert np.abs(e2.unet_forward(x, np.array([100.0], np.float32), ctx) - y1).max() > 1e-3


def test_taesd_file_and_vae_encoder_tensors_make_their_modules(sd, oracle, tmp_path):
    """A taesd checkpoint (names "decoder.layers.<i>. ...", loaded under the prefix "tae." like the reference's --taesd, stable-diffusion.cpp:798-803) and a VAE file that
    carries the encoder: the modules the engine makes on first use are made by the load, their tensors are filled (not reported unused) and the graphs use them."""
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    probe = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    probe.use_tae(True)
    probe.vae_encode(np.zeros((1, 3, 8, 8), np.float32))
    rng = np.random.default_rng(4)
    tae, enc = {}, {}
    for n in _names(probe):
        if not (n.startswith("tae.") or n.startswith("first_stage_model.encoder.") or n.startswith("first_stage_model.quant_conv.")):
            continue
        ne, ty, _ = probe.tensor_info(n)
        shape = tuple(int(d) for d in reversed(ne))
        while len(shape) > 1 and shape[0] == 1:
            shape = shape[1:]
        a = (rng.standard_normal(shape) * 0.05).astype(np.float16)
        (tae if n.startswith("tae.") else enc)[n[4:] if n.startswith("tae.") else n] = ("F16", a)
    assert not any(n.startswith("tae.") or ".encoder." in n for n in _names(e))
    _write_safetensors(tmp_path / "taesd.safetensors", tae)
    _write_safetensors(tmp_path / "vae_enc.safetensors", enc)
    r = e.load_weights(tmp_path / "taesd.safetensors", prefix="tae.")
    assert r["loaded"] == len(tae) and r["unused"] == 0
    r = e.load_weights(tmp_path / "vae_enc.safetensors")
    assert r["loaded"] == len(enc) and r["unused"] == 0
    np.testing.assert_array_equal(e.get_tensor("tae.decoder.layers.0.weight").ravel(), tae["decoder.layers.0.weight"][1].astype(np.float32).ravel())
    z = rng.standard_normal((1, 4, 4, 4)).astype(np.float32)
    assert np.abs(e.tae_decode(z) - probe.tae_decode(z)).max() > 1e-4
    img = rng.random((1, 3, 16, 16)).astype(np.float32)
    assert np.abs(e.vae_encode(img, return_moments=True)[1] - probe.vae_encode(img, return_moments=True)[1]).max() > 1e-4


def test_gguf_load_f16_f32_q8_0(sd, oracle, tmp_path):
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.Q8_0)
    rng = np.random.default_rng(4)
    picks = {
        "model.diffusion_model.input_blocks.0.0.weight": sd.F16,                                    # conv weight [3,3,4,32] f16
        "model.diffusion_model.input_blocks.0.0.bias": sd.F32,
        "model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight": sd.Q8_0,   # quantised Linear, stays q8_0 bit for bit
        "model.diffusion_model.time_embed.0.weight": sd.Q8_0,                                       # file q8_0 -> parameter f16 (never quantised)
    }
    tensors, want = [], {}
    for n, fty in picks.items():
        ne, pty, _ = e.tensor_info(n)
        ne = [int(d) for d in ne]
        while len(ne) > 1 and ne[-1] == 1:
            ne = ne[:-1]
        a = (rng.standard_normal(int(np.prod(ne))) * 0.1).astype(np.float32)
        if fty == sd.F32:
            raw, val = a.tobytes(), a
        elif fty == sd.F16:
            raw, val = a.astype(np.float16).tobytes(), a.astype(np.float16).astype(np.float32)
        else:
            raw, val = _q8_0_blocks(a)
        tensors.append((n, fty, ne, raw))
        want[n] = (val, pty, raw)
    p = tmp_path / "tiny.gguf"
    _write_gguf(p, tensors)
    r = e.load_weights(p)
    assert r["loaded"] == 4 and r["unused"] == 0 and r["missing"] > 100
    for n, (val, pty, raw) in want.items():
        got = e.get_tensor(n).ravel()
        if pty == sd.F16:
            val = val.astype(np.float16).astype(np.float32)
        np.testing.assert_allclose(got, val, rtol=0, atol=1e-7, err_msg=n)


def _q4_1_q5_blocks(a, kind):
    """Encoder for the test: f32 [n] (n % 32 == 0) -> raw q4_1 / q5_0 / q5_1 blocks (public ggml block layouts) and the values they decode to.
    q4_1 {f16 d, f16 m, qs[16]}, q5_0 {f16 d, qh[4], qs[16]}, q5_1 {f16 d, f16 m, qh[4], qs[16]}; byte j of qs = element j (low nibble) and
    j + 16 (high nibble); bit j of the little-endian u32 qh = fifth bit of element j."""
    x = a.reshape(-1, 32).astype(np.float32)
    levels = 15 if kind == "q4_1" else 31
    raw, vals = b"", []
    for row in x:
        if kind == "q5_0":
            amax = float(np.abs(row).max())
            d = np.float16(amax / 15.0 if amax > 0 else 1.0)
            q = np.clip(np.round(row / np.float32(d)) + 16, 0, 31).astype(np.int64)
            dec = (q - 16).astype(np.float32) * np.float32(d)
            head = np.array([d], np.float16).tobytes()
        else:
            lo, hi = float(row.min()), float(row.max())
            d = np.float16((hi - lo) / levels if hi > lo else 1.0)
            m = np.float16(lo)
            q = np.clip(np.round((row - np.float32(m)) / np.float32(d)), 0, levels).astype(np.int64)
            dec = q.astype(np.float32) * np.float32(d) + np.float32(m)
            head = np.array([d, m], np.float16).tobytes()
        qs = ((q[:16] & 0xF) | ((q[16:] & 0xF) << 4)).astype(np.uint8).tobytes()
        qh = b""
        if kind != "q4_1":
            bits = sum(int((q[j] >> 4) & 1) << j for j in range(32))
            qh = struct.pack("<I", bits)
        raw += head + qh + qs
        vals.append(dec)
    return raw, np.concatenate(vals)


@pytest.mark.parametrize("kind,gtype", [("q4_1", 3), ("q5_0", 6), ("q5_1", 7)])
def test_gguf_q4_1_q5_blocks_are_decoded_at_load(sd, oracle, tmp_path, kind, gtype):
    """GGUF checkpoints of the reference's users are often q4_1 / q5_0 / q5_1: the graph never computes in those types, the loader decodes them to
    f32 and re-encodes in the parameter's type like the reference's convert_tensor (src/model_loader.cpp:155-205)."""
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F16)
    rng = np.random.default_rng(gtype)
    names = ["model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight", "model.diffusion_model.time_embed.0.weight"]
    tensors, want = [], {}
    for n in names:
        ne, pty, _ = e.tensor_info(n)
        ne = [int(d) for d in ne]
        while len(ne) > 1 and ne[-1] == 1:
            ne = ne[:-1]
        assert ne[0] % 32 == 0
        a = (rng.standard_normal(int(np.prod(ne))) * 0.1).astype(np.float32)
        raw, val = _q4_1_q5_blocks(a, kind)
        tensors.append((n, gtype, ne, raw))
        want[n] = (val, pty)
    p = tmp_path / f"tiny_{kind}.gguf"
    _write_gguf(p, tensors)
    r = e.load_weights(p)
    assert r["loaded"] == len(names) and r["unused"] == 0
    for n, (val, pty) in want.items():
        got = e.get_tensor(n).ravel()
        if pty == sd.F16:
            val = val.astype(np.float16).astype(np.float32)
        np.testing.assert_allclose(got, val, rtol=0, atol=1e-7, err_msg=n)
    # a truncated block stream is rejected, not over-read
    bad = tmp_path / f"bad_{kind}.gguf"
    _write_gguf(bad, [(names[0], gtype, [64, 1 << 30], b"\0" * 44)])
    with pytest.raises(RuntimeError):
        e.load_weights(bad)


def _rand_f16(rng, n, lo=0.002, hi=0.05):
    return (rng.uniform(lo, hi, n) * rng.choice([-1.0, 1.0], n)).astype(np.float16)


def _kquant_random_blocks(rng, kind, nblk):
    """Random VALID super-blocks of a K-quant (every bit pattern of the integer fields is legal; the f16 scales are finite) and the values they decode
    to, computed here in vectorised numpy from the public block layouts (ggml-common.h) the way gguf-py's quants.py does — independent of the scalar
    loops in model_io.hpp, which follow ggml-quants.c dequantize_row_q*_K."""
    if kind == "q2_K":  # {u8 scales[16]; u8 qs[64]; f16 d; f16 dmin}
        scales = rng.integers(0, 256, (nblk, 16), dtype=np.uint8)
        qs = rng.integers(0, 256, (nblk, 64), dtype=np.uint8)
        d, dmin = _rand_f16(rng, nblk), _rand_f16(rng, nblk)
        raw = b"".join(scales[i].tobytes() + qs[i].tobytes() + d[i].tobytes() + dmin[i].tobytes() for i in range(nblk))
        dl = d.astype(np.float32)[:, None] * (scales & 0xF).astype(np.float32)          # [nblk, 16]
        ml = dmin.astype(np.float32)[:, None] * (scales >> 4).astype(np.float32)
        q = (qs.reshape(nblk, 2, 1, 32) >> np.array([0, 2, 4, 6], np.uint8).reshape(1, 1, 4, 1)) & 3   # [nblk, half, shift, 32]
        q = q.reshape(nblk, 16, 16).astype(np.float32)                                   # 16 groups of 16 in output order
        val = dl[:, :, None] * q - ml[:, :, None]
    elif kind == "q3_K":  # {u8 hmask[32]; u8 qs[64]; u8 scales[12]; f16 d}
        hmask = rng.integers(0, 256, (nblk, 32), dtype=np.uint8)
        qs = rng.integers(0, 256, (nblk, 64), dtype=np.uint8)
        sc12 = rng.integers(0, 256, (nblk, 12), dtype=np.uint8)
        d = _rand_f16(rng, nblk)
        raw = b"".join(hmask[i].tobytes() + qs[i].tobytes() + sc12[i].tobytes() + d[i].tobytes() for i in range(nblk))
        lo = np.concatenate([sc12[:, :8] & 0xF, sc12[:, :8] >> 4], axis=1)               # 16 low nibbles
        hi = np.stack([(sc12[:, 8:12] >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(nblk, 16)
        scales = (lo | (hi << 4)).astype(np.int32) - 32
        q = (qs.reshape(nblk, 2, 1, 32) >> np.array([0, 2, 4, 6], np.uint8).reshape(1, 1, 4, 1)) & 3
        q = q.reshape(nblk, 8, 32).astype(np.int32)                                      # 8 runs of 32 weights
        hb = ((hmask[:, None, :] >> np.arange(8, dtype=np.uint8).reshape(1, 8, 1)) & 1).astype(np.int32)
        q = q - 4 * (1 - hb)
        val = d.astype(np.float32)[:, None, None] * scales.reshape(nblk, 16, 1).astype(np.float32) * q.reshape(nblk, 16, 16).astype(np.float32)
    elif kind in ("q4_K", "q5_K"):  # {f16 d; f16 dmin; u8 scales[12]; [u8 qh[32];] u8 qs[128]}
        sc12 = rng.integers(0, 256, (nblk, 12), dtype=np.uint8)
        qs = rng.integers(0, 256, (nblk, 128), dtype=np.uint8)
        qh = rng.integers(0, 256, (nblk, 32), dtype=np.uint8)
        d, dmin = _rand_f16(rng, nblk), _rand_f16(rng, nblk)
        raw = b"".join(d[i].tobytes() + dmin[i].tobytes() + sc12[i].tobytes() + (qh[i].tobytes() if kind == "q5_K" else b"") + qs[i].tobytes() for i in range(nblk))
        sc = np.concatenate([sc12[:, 0:4] & 63, (sc12[:, 8:12] & 0xF) | ((sc12[:, 0:4] >> 6) << 4)], axis=1).astype(np.float32)
        mn = np.concatenate([sc12[:, 4:8] & 63, (sc12[:, 8:12] >> 4) | ((sc12[:, 4:8] >> 6) << 4)], axis=1).astype(np.float32)
        q4 = qs.reshape(nblk, 4, 32)
        q = np.stack([q4 & 0xF, q4 >> 4], axis=2).reshape(nblk, 8, 32).astype(np.int32)  # group g: low nibbles then high nibbles
        if kind == "q5_K":
            q = q + 16 * ((qh[:, None, :] >> np.arange(8, dtype=np.uint8).reshape(1, 8, 1)) & 1).astype(np.int32)
        val = d.astype(np.float32)[:, None, None] * sc[:, :, None] * q.astype(np.float32) - dmin.astype(np.float32)[:, None, None] * mn[:, :, None]
    elif kind == "q6_K":  # {u8 ql[128]; u8 qh[64]; i8 scales[16]; f16 d}
        ql = rng.integers(0, 256, (nblk, 128), dtype=np.uint8)
        qh = rng.integers(0, 256, (nblk, 64), dtype=np.uint8)
        sc = rng.integers(-128, 128, (nblk, 16), dtype=np.int8)
        d = _rand_f16(rng, nblk, 0.0005, 0.004)
        raw = b"".join(ql[i].tobytes() + qh[i].tobytes() + sc[i].tobytes() + d[i].tobytes() for i in range(nblk))
        l2 = ql.reshape(nblk, 2, 2, 32)                                                  # [half][l32 block 0/1][l]
        lo = np.stack([l2[:, :, 0] & 0xF, l2[:, :, 1] & 0xF, l2[:, :, 0] >> 4, l2[:, :, 1] >> 4], axis=2)  # [nblk, half, c, 32]
        h2 = qh.reshape(nblk, 2, 1, 32)
        hi = (h2 >> np.array([0, 2, 4, 6], np.uint8).reshape(1, 1, 4, 1)) & 3
        q = (lo | (hi << 4)).astype(np.int32) - 32                                       # weight half*128 + 32 c + l
        s = sc.reshape(nblk, 2, 4, 2).astype(np.float32)                                 # scale index half*8 + 2 c + l // 16
        s = np.repeat(s, 16, axis=3)                                                     # [nblk, half, c, 32]
        val = (d.astype(np.float32)[:, None, None, None] * s) * q.astype(np.float32)
    else:  # iq4_nl {f16 d; u8 qs[16]}
        kv = np.array([-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113], np.float32)
        qs = rng.integers(0, 256, (nblk, 16), dtype=np.uint8)
        d = _rand_f16(rng, nblk, 0.0002, 0.002)
        raw = b"".join(d[i].tobytes() + qs[i].tobytes() for i in range(nblk))
        val = d.astype(np.float32)[:, None] * np.concatenate([kv[qs & 0xF], kv[qs >> 4]], axis=1)
    return raw, val.astype(np.float32).reshape(-1)


@pytest.mark.parametrize("kind,gtype,blck,bytes_", [("q2_K", 10, 256, 84), ("q3_K", 11, 256, 110), ("q4_K", 12, 256, 144), ("q5_K", 13, 256, 176),
                                                     ("q6_K", 14, 256, 210), ("iq4_nl", 20, 32, 18)])
def test_gguf_k_quants_are_decoded_at_load(sd, oracle, tmp_path, kind, gtype, blck, bytes_):
    """The q2_k / q3_k / q4_k / q5_k / q6_k GGUF files the reference's docs point FLUX / SD3.5 users at (docs/flux.md:36-38): 256-weight super-blocks,
    decoded to f32 at load and re-encoded in the parameter's type like convert_tensor does for every source type (src/model_loader.cpp:155-205)."""
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F16)
    rng = np.random.default_rng(100 + gtype)
    names = ["model.diffusion_model.input_blocks.4.1.transformer_blocks.0.ff.net.2.weight",   # [256, 64]  f16 parameter
             "model.diffusion_model.input_blocks.7.1.transformer_blocks.0.ff.net.0.proj.bias"]  # [1024]     f32 parameter
    tensors, want = [], {}
    for n in names:
        ne, pty, _ = e.tensor_info(n)
        ne = [int(d) for d in ne]
        while len(ne) > 1 and ne[-1] == 1:
            ne = ne[:-1]
        assert ne[0] % blck == 0
        raw, val = _kquant_random_blocks(rng, kind, int(np.prod(ne)) // blck)
        assert len(raw) == int(np.prod(ne)) // blck * bytes_
        tensors.append((n, gtype, ne, raw))
        want[n] = (val, pty)
    p = tmp_path / f"tiny_{kind}.gguf"
    _write_gguf(p, tensors)
    r = e.load_weights(p)
    assert r["loaded"] == len(names) and r["unused"] == 0
    for n, (val, pty) in want.items():
        got = e.get_tensor(n).ravel()
        if pty == sd.F16:
            val = val.astype(np.float16).astype(np.float32)
        # f32 parameters: bit-equal up to the last ulp of the two multiplication orders; f16 parameters: a value on a rounding tie may land one f16 step away
        np.testing.assert_allclose(got, val, rtol=1e-3 if pty == sd.F16 else 2e-7, atol=1e-7, err_msg=n)
    # a truncated block stream is rejected, not over-read; a row that is not a whole number of blocks too
    _write_gguf(tmp_path / "bad1.gguf", [(names[0], gtype, [256, 1 << 30], b"\0" * 300)])
    with pytest.raises(RuntimeError):
        e.load_weights(tmp_path / "bad1.gguf")
    _write_gguf(tmp_path / "bad2.gguf", [(names[0], gtype, [blck + 8, 64], b"\0" * (bytes_ * 80))])
    with pytest.raises(RuntimeError):
        e.load_weights(tmp_path / "bad2.gguf")


def test_bad_files_are_rejected(sd, oracle, tmp_path):
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    p = tmp_path / "junk.bin"
    p.write_bytes(b"\x05\x00\x00\x00\x00\x00\x00\x00hello")
    with pytest.raises(sd.EngineError):
        e.load_weights(p)
    with pytest.raises(sd.EngineError):
        e.load_weights(tmp_path / "missing.safetensors")
    # right name, wrong shape
    _write_safetensors(tmp_path / "bad.safetensors", {"model.diffusion_model.input_blocks.0.0.bias": ("F32", np.zeros((7,), np.float32))})
    with pytest.raises(sd.EngineError):
        e.load_weights(tmp_path / "bad.safetensors")


def _e4m3_decode_np(b):
    """OCP FP8 E4M3 ("fn") from the format definition: bias 7, no infinities, S.1111.111 = NaN"""
    b = b.astype(np.int32)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e == 0, m / 8.0 * 2.0**-6, (1 + m / 8.0) * 2.0 ** (e - 7.0))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return np.where(s == 1, -v, v).astype(np.float32)


def test_safetensors_5d_tensors_fold_their_outer_dims(sd, oracle, tmp_path):
    """safetensors_io.cpp:265-283: a 5-D tensor is accepted with its two outermost (torch-order) dims folded into one; more than five
    dims is the reference's 'invalid tensor' error.  A 4-D conv weight written as [2, OC/2, IC, KH, KW] must load as the same bytes, and an
    undeclared 5-D tensor (a bundled 3-D conv) must not stop the file from loading."""
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    names = _names(e)
    rng = np.random.default_rng(11)
    tensors, want, folded = {}, {}, 0
    for n in names:
        ne, ty, _ = e.tensor_info(n)
        shape = tuple(int(d) for d in reversed(ne))
        while len(shape) > 1 and shape[0] == 1:
            shape = shape[1:]
        a = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        want[n] = a
        if len(shape) == 4 and shape[0] % 2 == 0 and folded < 6:
            a = a.reshape((2, shape[0] // 2) + shape[1:])
            folded += 1
        tensors[n] = ("F32", a)
    assert folded == 6
    tensors["bundled.conv3d.weight"] = ("F32", np.zeros((2, 3, 2, 2, 2), np.float32))
    p = tmp_path / "tiny5d.safetensors"
    _write_safetensors(p, tensors)
    assert e.load_weights(p) == {"loaded": len(names), "missing": 0, "unused": 1}
    for n in names[::9]:
        _, ty, _ = e.tensor_info(n)
        ref = want[n].ravel()
        if ty == sd.F16:
            ref = ref.astype(np.float16).astype(np.float32)
        np.testing.assert_array_equal(e.get_tensor(n).ravel(), ref, err_msg=n)
    tensors["six.dims"] = ("F32", np.zeros((1, 1, 2, 1, 1, 2), np.float32))
    p6 = tmp_path / "tiny6d.safetensors"
    _write_safetensors(p6, tensors)
    with pytest.raises(Exception, match="invalid tensor"):
        e.load_weights(p6)


def test_safetensors_wide_and_8bit_dtypes_are_converted(sd, oracle, tmp_path):
    """F64 / I64 / F8_E4M3 / F8_E5M2 payloads are widened to f32 and then converted to the parameter's type, as the reference does
    (safetensors_io.cpp:79-99, model_loader.cpp:81-153) — round-1 advice: they used to be dropped silently."""
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    rng = np.random.default_rng(5)
    nb, nw = "model.diffusion_model.input_blocks.0.0.bias", "model.diffusion_model.out.2.bias"
    n8, n52 = "model.diffusion_model.time_embed.0.bias", "model.diffusion_model.time_embed.2.bias"
    shp = lambda n: tuple(int(d) for d in reversed(e.tensor_info(n)[0]) if d != 1) or (1,)
    a64 = rng.standard_normal(shp(nb))
    i64 = rng.integers(-5, 5, shp(nw)).astype(np.int64)
    b8 = rng.integers(0, 256, shp(n8)).astype(np.uint8)
    b8[(b8 & 0x7F) == 0x7F] = 0x3C   # keep NaNs out of the weights
    b52 = rng.integers(0, 0x7B, shp(n52)).astype(np.uint8)   # below the E5M2 inf / NaN codes
    _write_safetensors(tmp_path / "wide.safetensors", {nb: ("F64", a64), nw: ("I64", i64), n8: ("F8_E4M3", b8), n52: ("F8_E5M2", b52)})
    r = e.load_weights(tmp_path / "wide.safetensors")
    assert r["loaded"] == 4 and r["unused"] == 0
    np.testing.assert_array_equal(e.get_tensor(nb).ravel(), a64.astype(np.float32).ravel())
    np.testing.assert_array_equal(e.get_tensor(nw).ravel(), i64.astype(np.float32).ravel())
    np.testing.assert_array_equal(e.get_tensor(n8).ravel(), _e4m3_decode_np(b8).ravel())
    np.testing.assert_array_equal(e.get_tensor(n52).ravel(), (b52.astype(np.uint16) << 8).view(np.float16).astype(np.float32).ravel())


def test_malformed_headers_are_rejected_not_overread(sd, oracle, tmp_path):
    """Round-1 advice: a byte range shorter than prod(shape) * type_size, an undecodable dtype of a declared parameter, and GGUF dims that are
    negative / not whole blocks / overflowing must be errors (they used to become heap over-reads or silent synthetic weights)."""
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle)
    name = "model.diffusion_model.input_blocks.0.0.bias"
    n = int(np.prod(e.tensor_info(name)[0]))
    # 1. data_offsets shorter than the shape needs
    hdr = json.dumps({name: {"dtype": "F32", "shape": [n], "data_offsets": [0, 4 * n - 8]}}).encode()
    (tmp_path / "short.safetensors").write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * (4 * n))
    with pytest.raises(sd.EngineError, match="size mismatch"):
        e.load_weights(tmp_path / "short.safetensors")
    # 2. a dtype no reader decodes, on a tensor the model declares
    _write_safetensors(tmp_path / "i8.safetensors", {name: ("I8", np.zeros((n,), np.int8))})
    with pytest.raises(sd.EngineError, match="cannot decode"):
        e.load_weights(tmp_path / "i8.safetensors")
    # ... while an undecodable tensor the model does NOT declare is just unused
    _write_safetensors(tmp_path / "i8b.safetensors", {"unrelated.flag": ("BOOL", np.zeros((3,), np.uint8)), name: ("F32", np.ones((n,), np.float32))})
    assert e.load_weights(tmp_path / "i8b.safetensors")["loaded"] == 1
    # 3. GGUF: a type this build does not decode (iq2_xxs = 16; the K-quants and iq4_nl ARE decoded) on a declared tensor; ne0 not a whole number
    #    of blocks; a dimension that overflows; data past the end of the file
    wname = "model.diffusion_model.time_embed.0.weight"
    ne = [int(d) for d in e.tensor_info(wname)[0][:2]]
    _write_gguf(tmp_path / "q4k.gguf", [(wname, 16, ne, b"\0" * 64)])
    with pytest.raises(sd.EngineError, match="cannot decode"):
        e.load_weights(tmp_path / "q4k.gguf")
    for bad_ne, raw in (([33, 4], b"\0" * 34 * 8), ([32, 2**62], b"\0" * 34), ([32, 2**20], b"\0" * 34)):
        _write_gguf(tmp_path / "bad.gguf", [(wname, sd.Q8_0, bad_ne, raw)])
        with pytest.raises(sd.EngineError, match="invalid dimensions|outside the file"):
            e.load_weights(tmp_path / "bad.gguf")


# ---- PyTorch checkpoints (.ckpt / .pt): files written by torch.save itself — the format's own reference implementation ------------------------------
def _tiny_state(e, rng, names):
    """{name: np array in torch order} for the given parameters of the tiny engine"""
    out = {}
    for n in names:
        ne, _, _ = e.tensor_info(n)
        shape = [int(d) for d in ne]
        while len(shape) > 1 and shape[-1] == 1:
            shape = shape[:-1]
        out[n] = (rng.standard_normal(shape[::-1]) * 0.1).astype(np.float32)
    return out


_CKPT_NAMES = ["model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight", "model.diffusion_model.time_embed.0.weight",
               "model.diffusion_model.input_blocks.0.0.weight", "model.diffusion_model.input_blocks.0.0.bias", "model.diffusion_model.out.0.weight"]


@pytest.mark.parametrize("legacy", [False, True])
@pytest.mark.parametrize("wrap", ["bare", "lightning"])
def test_torch_checkpoint_load(sd, oracle, tmp_path, legacy, wrap):
    """torch.save'd checkpoints (the zip container of PyTorch >= 1.6 and the legacy stream, src/model_io/torch_zip_io.cpp / torch_legacy_io.cpp /
    pickle_io.cpp): a bare state dict and a pytorch-lightning style {"epoch", "global_step", "state_dict": OrderedDict, "optimizer_states": [...]}
    with f32 / f16 / bf16 / f64 tensors, a view into a shared storage (non-zero storage offset), an int64 step counter the model does not
    declare, and an optimizer state holding tensors under names that are not parameters.  Loaded through sd_load_weights and compared tensor by tensor."""
    import collections
    import torch
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F16)
    rng = np.random.default_rng(7)
    st = _tiny_state(e, rng, _CKPT_NAMES)
    sdict = collections.OrderedDict()
    dts = [torch.float32, torch.float16, torch.bfloat16, torch.float64, torch.float32]
    want = {}
    for (n, a), dt in zip(st.items(), dts):
        t = torch.from_numpy(a).to(dt)
        sdict[n] = t
        want[n] = t.to(torch.float32).numpy()
    # a view: two parameters over ONE storage, the second at a non-zero element offset
    n0, n1 = _CKPT_NAMES[3], "model.diffusion_model.input_blocks.1.0.in_layers.0.bias"
    c0 = st[n0].size
    c1 = int(np.prod([int(d) for d in e.tensor_info(n1)[0]]))
    shared = torch.from_numpy((rng.standard_normal(c0 + c1) * 0.1).astype(np.float32))
    sdict[n0], sdict[n1] = shared[:c0], shared[c0:]
    want[n0], want[n1] = shared[:c0].numpy(), shared[c0:].numpy()
    sdict["model.diffusion_model.some_counter"] = torch.tensor([3], dtype=torch.int64)  # not a parameter: ignored
    obj = sdict
    if wrap == "lightning":
        obj = {"epoch": 6, "global_step": 470000, "pytorch-lightning_version": "1.4.2", "state_dict": sdict, "lr": 1e-4, "flag": True, "none": None,
               "optimizer_states": [{"state": {0: {"exp_avg": torch.zeros(3, 5), "step": 17}}, "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.999)}]}],
               "callbacks": {"ckpt": {"best": torch.tensor(0.25)}}}
    p = tmp_path / ("m_legacy.ckpt" if legacy else "m.ckpt")
    torch.save(obj, p, _use_new_zipfile_serialization=not legacy)
    r = e.load_weights(p)
    assert r["loaded"] == len(want), r
    for n, val in want.items():
        got = e.get_tensor(n)
        _, pty, _ = e.tensor_info(n)
        ref = val.reshape(got.shape)
        if pty == sd.F16:
            ref = ref.astype(np.float16).astype(np.float32)
        np.testing.assert_array_equal(got, ref, err_msg=n)


def test_torch_checkpoint_rejects_what_it_cannot_read(sd, oracle, tmp_path):
    """Non-contiguous tensors are not offered to the loader (a declared parameter stored transposed is reported missing, never read with the wrong
    layout); an integer storage on a declared parameter is an error; truncated archives and pickles are rejected, not over-read."""
    import torch
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F16)
    rng = np.random.default_rng(8)
    name = _CKPT_NAMES[1]
    a = _tiny_state(e, rng, [name])[name]
    good = tmp_path / "good.ckpt"
    torch.save({name: torch.from_numpy(a)}, good)
    assert e.load_weights(good)["loaded"] == 1
    # transposed view: same storage, strides swapped
    tt = torch.from_numpy(np.ascontiguousarray(a.T)).t()
    assert tt.shape == a.shape and not tt.is_contiguous()
    torch.save({name: tt}, tmp_path / "nc.ckpt")
    with pytest.raises(sd.EngineError):
        e.load_weights(tmp_path / "nc.ckpt")  # nothing loadable in the file
    torch.save({name: torch.from_numpy(a).to(torch.int32)}, tmp_path / "int.ckpt")
    with pytest.raises(sd.EngineError, match="cannot decode"):
        e.load_weights(tmp_path / "int.ckpt")
    raw = good.read_bytes()
    for cut in (len(raw) - 10, len(raw) // 2, 40):
        (tmp_path / "cut.ckpt").write_bytes(raw[:cut])
        with pytest.raises(sd.EngineError):
            e.load_weights(tmp_path / "cut.ckpt")
    # a storage entry shorter than the pickle says: corrupt the element count of the shape inside data.pkl is hard to do portably — instead point the
    # central directory's size of the storage entry below the tensor's byte count
    legacy = tmp_path / "leg.ckpt"
    torch.save({name: torch.from_numpy(a)}, legacy, _use_new_zipfile_serialization=False)
    lraw = legacy.read_bytes()
    (tmp_path / "legcut.ckpt").write_bytes(lraw[: len(lraw) - a.nbytes // 2])
    with pytest.raises(sd.EngineError):
        e.load_weights(tmp_path / "legcut.ckpt")


@pytest.mark.parametrize("legacy", [False, True])
def test_torch_checkpoint_with_a_zero_element_tensor_loads(sd, oracle, tmp_path, legacy):
    """Round-4 advice: torch.save({..., 'empty': torch.zeros(0, 4)}) — the reference accepts such files (pickle_io.cpp: has_zero_dimension);
    a zero-element tensor next to good ones must not abort the load (it carries no data: listed as unused)."""
    import torch
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F16)
    rng = np.random.default_rng(9)
    names = _CKPT_NAMES[:2]
    st = _tiny_state(e, rng, names)
    obj = {"state_dict": {**{k: torch.from_numpy(v) for k, v in st.items()}, "empty": torch.zeros(0, 4), "also.empty": torch.zeros(3, 0, 2)}}
    p = tmp_path / "z.ckpt"
    torch.save(obj, p, _use_new_zipfile_serialization=not legacy)
    r = e.load_weights(p)
    assert r["loaded"] == len(names), r
    for n, val in st.items():
        got = e.get_tensor(n)
        ref = val.reshape(got.shape)
        if e.tensor_info(n)[1] == sd.F16:
            ref = ref.astype(np.float16).astype(np.float32)
        np.testing.assert_array_equal(got, ref, err_msg=n)
