"""Lane-level restatements (NumPy) of two pieces of device logic that are easy to get wrong and cannot run on the CPU box:
  * k_qgemm16's in-register dequantisation (csrc/kernels/qgemm.hip): which strip bytes a lane of v_mfma_f32_32x32x16_f16 picks for its B fragment,
    the 2-byte-phase realignment, the v_perm / 0x6400 integer->f16 trick and the k order inside a block that the A fragment has to follow;
  * wave_sum_transpose: the halving cross-lane reduction of the few-row kernels and which lane ends up owning which value.
They pin the index arithmetic the GPU parity tests (tests/test_gpu_ops.py::test_quantised_mfma_gemm_raw_blocks, test_few_row_linear_weight_stream)
then confirm on hardware."""
import numpy as np
import pytest


def v_perm(a: int, b: int, sel: int) -> int:
    """v_perm_b32 D = bytes of {a (4..7), b (0..3)} picked by the selector bytes"""
    src = [(b >> (8 * i)) & 0xFF for i in range(4)] + [(a >> (8 * i)) & 0xFF for i in range(4)]
    return sum(src[(sel >> (8 * i)) & 0xFF] << (8 * i) for i in range(4))


def halfs(u: int) -> np.ndarray:
    return np.array([u & 0xFFFF, (u >> 16) & 0xFFFF], dtype=np.uint16).view(np.float16)


def deq4(u: int, off: float, d: np.float16) -> list:
    """qg_deq4: four bytes -> four f16 weights (byte + 1024 via the exponent byte 0x64, minus off, times the block scale)"""
    lo = (halfs(v_perm(0x64646464, u, 0x04010400)) - np.float16(off)).astype(np.float16) * d
    hi = (halfs(v_perm(0x64646464, u, 0x04030402)) - np.float16(off)).astype(np.float16) * d
    return list(lo.astype(np.float16)) + list(hi.astype(np.float16))


@pytest.mark.parametrize("qt", [8, 4])
def test_qgemm16_fragment_bytes_and_k_order(qt):
    rng = np.random.default_rng(qt)
    BLK, SEG = (34, 8) if qt == 8 else (18, 8)
    CS, NB = SEG * BLK, (16 if qt == 8 else 8)
    strip = np.zeros(32 * CS, dtype=np.uint8)          # a wave's LDS strip: [column][8 raw blocks]
    wref = np.zeros((32, 256), dtype=np.float32)       # f16(d * q): what the f16 weight image would hold
    for n in range(32):
        for b in range(SEG):
            d = np.float16(rng.uniform(0.01, 0.1))
            o = n * CS + BLK * b
            strip[o:o + 2] = np.array([d], dtype=np.float16).view(np.uint8)
            if qt == 8:
                q = rng.integers(-128, 128, 32).astype(np.int8)
                strip[o + 2:o + 34] = q.view(np.uint8)
                wref[n, 32 * b:32 * b + 32] = (np.float32(d) * q.astype(np.float32)).astype(np.float16)
            else:
                nib = rng.integers(0, 16, 32).astype(np.uint8)
                strip[o + 2:o + 18] = nib[:16] | (nib[16:] << 4)     # ggml q4_0: byte j = element j (low nibble) and j + 16 (high nibble)
                wref[n, 32 * b:32 * b + 32] = (np.float32(d) * (nib.astype(np.float32) - 8)).astype(np.float16)

    def rd32(off):
        assert off % 4 == 0, "LDS dword reads must be aligned"
        return int(strip[off:off + 4].view(np.uint32)[0])

    A = rng.standard_normal((32, 256)).astype(np.float16)
    D = np.zeros((32, 32))
    for b in range(SEG):
        ph = (BLK * b + 2) & 2                         # compile-time phase of the quant bytes inside a dword
        for lane in range(64):
            n, kg = lane & 31, lane >> 5
            blk = n * CS + BLK * b
            d = strip[blk:blk + 2].view(np.float16)[0]
            qp = blk + 2 + NB * kg
            if ph == 0:
                q = [rd32(qp + 4 * j) for j in range(NB // 4)]
            else:
                raw = [rd32(qp - 2 + 4 * j) for j in range(NB // 4 + 1)]
                q = [(((raw[j + 1] << 32) | raw[j]) >> 16) & 0xFFFFFFFF for j in range(NB // 4)]   # v_alignbit_b32(hi, lo, 16)
            if qt == 8:
                f0 = deq4(q[0] ^ 0x80808080, 1152, d) + deq4(q[1] ^ 0x80808080, 1152, d)
                f1 = deq4(q[2] ^ 0x80808080, 1152, d) + deq4(q[3] ^ 0x80808080, 1152, d)
                k0, k1 = 32 * b + 16 * kg, 32 * b + 16 * kg + 8
            else:
                f0 = deq4(q[0] & 0x0F0F0F0F, 1032, d) + deq4(q[1] & 0x0F0F0F0F, 1032, d)
                f1 = deq4((q[0] >> 4) & 0x0F0F0F0F, 1032, d) + deq4((q[1] >> 4) & 0x0F0F0F0F, 1032, d)
                k0, k1 = 32 * b + 8 * kg, 32 * b + 16 + 8 * kg
            for f, k in ((f0, k0), (f1, k1)):
                f = np.array(f, dtype=np.float32)
                np.testing.assert_array_equal(f, wref[n, k:k + 8])       # bit-identical to the image's f16 value
                D[:, n] += A[:, k:k + 8].astype(np.float64) @ f          # the MFMA step: A fragment read at the same k offsets
    np.testing.assert_allclose(D, A.astype(np.float64) @ wref.astype(np.float64).T, rtol=0, atol=1e-9)


@pytest.mark.parametrize("nv", [1, 2, 4, 8, 16, 32])
def test_wave_sum_transpose_lane_ownership(nv):
    rng = np.random.default_rng(nv)
    log = int(np.log2(nv))
    v = rng.standard_normal((64, nv))                  # v[lane][value]
    orig = v.copy()
    lanes = np.arange(64)
    for st in range(log):
        n, o = nv >> st, 32 >> st
        up = (lanes & o) != 0
        new = v.copy()
        for i in range(n // 2):
            send = np.where(up, v[:, i], v[:, i + n // 2])
            keep = np.where(up, v[:, i + n // 2], v[:, i])
            new[:, i] = keep + send[lanes ^ o]         # __shfl_xor(send, o)
        v = new
    r = v[:, 0].copy()
    o = 32 >> log
    while o > 0:
        r = r + r[lanes ^ o]
        o >>= 1
    grp = 64 // nv
    tot = orig.sum(0)
    for lane in range(64):
        assert abs(r[lane] - tot[lane // grp]) < 1e-9  # value index (lane / GRP) is complete in every lane of its group
