"""Lane-level restatements (NumPy) of pieces of device logic that are easy to get wrong and cannot run on the CPU box:
  * k_qgemm16's in-register dequantisation (csrc/kernels/qgemm.hip): which strip bytes a lane of v_mfma_f32_32x32x16_f16 picks for its B fragment,
    the 2-byte-phase realignment, the v_perm / 0x6400 integer->f16 trick and the k order inside a block that the A fragment has to follow;
  * wave_sum_transpose: the halving cross-lane reduction of the few-row kernels and which lane ends up owning which value.
  * k_wswz_q: the just-in-time f16 weight image built from raw q8_0 / q4_0 blocks (which bytes a lane turns into which fragment);
  * k_joint_heads: the DiT attention-operand pass (group / lane indexing, butterfly RMSNorm, rotary pairs, head-major destination).
  * round 4: epi_geglu16's value / gate pairing by v_permlane16_swap on the 16-column-interleaved FF1 weight image; gn_ld4's float4 indexing of a never-built
    channel concatenation; the unit ranges, part owners and slab slots of the stream-K kernel.
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


@pytest.mark.parametrize("qt", [8, 4])
def test_wswz_q_image_fragments_in_natural_k_order(qt):
    """k_wswz_q (csrc/kernels/qgemm.hip): the just-in-time weight image built from raw q8_0 / q4_0 blocks.  The image is k_gemm16's operand
    [rows/32][K/16][64 lanes][8 halfs] in NATURAL k order: lane (n = lane % 32, hi = lane / 32) of fragment kb holds W[rb*32 + n][kb*16 + hi*8 .. +8].
    Per block b of a 256-k segment a lane writes fragment 2b from the low byte run / low nibbles and fragment 2b+1 from the high run / high nibbles:
    this pins which raw bytes that is (it differs from k_qgemm16's permuted in-block order for q8_0)."""
    rng = np.random.default_rng(100 + qt)
    BLK, SEG = (34, 8) if qt == 8 else (18, 8)
    CS = SEG * BLK
    strip = np.zeros(32 * CS, dtype=np.uint8)
    wref = np.zeros((32, 256), dtype=np.float32)
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
                strip[o + 2:o + 18] = nib[:16] | (nib[16:] << 4)
                wref[n, 32 * b:32 * b + 32] = (np.float32(d) * (nib.astype(np.float32) - 8)).astype(np.float16)

    def rd32(off):
        assert off % 4 == 0, "LDS dword reads must be aligned"
        return int(strip[off:off + 4].view(np.uint32)[0])

    def rd8(qp, ph):   # eight bytes at qp: two dwords, realigned with v_alignbit when the run starts 2 bytes into a dword
        if ph == 0:
            return [rd32(qp), rd32(qp + 4)]
        r = [rd32(qp - 2), rd32(qp + 2), rd32(qp + 6)]
        return [(((r[1] << 32) | r[0]) >> 16) & 0xFFFFFFFF, (((r[2] << 32) | r[1]) >> 16) & 0xFFFFFFFF]

    image = np.zeros((16, 64, 8), dtype=np.float32)   # the 16 fragments of this segment for one 32-row block
    for b in range(SEG):
        ph = (BLK * b + 2) & 2
        for lane in range(64):
            n, hi = lane & 31, lane >> 5
            blk = n * CS + BLK * b
            d = strip[blk:blk + 2].view(np.float16)[0]
            if qt == 8:
                qa, qb = rd8(blk + 2 + 8 * hi, ph), rd8(blk + 2 + 16 + 8 * hi, ph)
                f0 = deq4(qa[0] ^ 0x80808080, 1152, d) + deq4(qa[1] ^ 0x80808080, 1152, d)
                f1 = deq4(qb[0] ^ 0x80808080, 1152, d) + deq4(qb[1] ^ 0x80808080, 1152, d)
            else:
                qa = rd8(blk + 2 + 8 * hi, ph)
                f0 = deq4(qa[0] & 0x0F0F0F0F, 1032, d) + deq4(qa[1] & 0x0F0F0F0F, 1032, d)
                f1 = deq4((qa[0] >> 4) & 0x0F0F0F0F, 1032, d) + deq4((qa[1] >> 4) & 0x0F0F0F0F, 1032, d)
            image[2 * b, lane] = np.array(f0, dtype=np.float32)
            image[2 * b + 1, lane] = np.array(f1, dtype=np.float32)
    # the image definition of wgemm.hip (k_wswz_linear): fragment kb, lane -> W[n][kb * 16 + hi * 8 + j]
    for kb in range(16):
        for lane in range(64):
            n, hi = lane & 31, lane >> 5
            np.testing.assert_array_equal(image[kb, lane], wref[n, kb * 16 + hi * 8:kb * 16 + hi * 8 + 8])


@pytest.mark.parametrize("d,H,La,Lb,N,norm,rope", [(64, 3, 5, 9, 2, True, False), (128, 2, 4, 6, 1, True, True), (64, 2, 0 + 7, 0, 2, False, False), (128, 1, 3, 5, 2, False, True)])
def test_joint_heads_indexing_norm_and_rotary(d, H, La, Lb, N, norm, rope):
    """k_joint_heads (csrc/kernels/elementwise.hip): group gi of d/4 lanes handles head h = gi % H of joint token l = (gi / H) % Lt of image
    n = gi / (H Lt); lane j owns elements 4j .. 4j+3; source row = the stream's projection row (n L + l') at column h d of the q / k / v
    slice; per-head RMSNorm through a xor butterfly over the group; rotary on the pairs (4j, 4j+1), (4j+2, 4j+3) with the table entry of
    joint position l; destination [d, Lt, H, N] head-major.  Checked against the node-by-node definition (split / norm / concat / rope / permute)."""
    rng = np.random.default_rng(d + H)
    C, Lt, G = d * H, La + Lb, d // 4
    xs_a, xs_b = 3 * C, 3 * C + 64          # row strides of the two projections (the second one as in FLUX's linear1: extra columns)
    col = C                                  # the k slice
    Ta = rng.standard_normal((N * max(La, 1), xs_a)).astype(np.float32)
    Tb = rng.standard_normal((N * max(Lb, 1), xs_b)).astype(np.float32)
    wa = rng.standard_normal(d).astype(np.float32)
    wb = rng.standard_normal(d).astype(np.float32)
    eps = 1e-6
    ang = rng.uniform(0, 6.28, (Lt, d // 2))
    pe = np.stack([np.stack([np.cos(ang), -np.sin(ang)], -1), np.stack([np.sin(ang), np.cos(ang)], -1)], -2).astype(np.float32)   # [Lt, d/2, 2, 2]
    out = np.zeros((N, H, Lt, d), dtype=np.float32)
    for gi in range(H * Lt * N):
        h, t = gi % H, gi // H
        l, n = t % Lt, t // Lt
        first = l < La
        row = Ta[n * La + l] if first else Tb[n * Lb + (l - La)]
        v = row[col + h * d:col + (h + 1) * d].copy().reshape(G, 4)           # lane j holds v[j]
        w = wa if first else wb
        if norm:
            ss = (v * v).sum(1)                                                 # per-lane partial
            m = G // 2
            while m >= 1:                                                        # xor butterfly: every lane ends with the group total
                ss = ss + ss[np.arange(G) ^ m]
                m //= 2
            sc = np.float32(1.0) / np.sqrt(ss / np.float32(d) + np.float32(eps))
            v = v * sc[:, None] * w.reshape(G, 4)
        if rope:
            for j in range(G):
                m0, m1 = pe[l, 2 * j], pe[l, 2 * j + 1]
                x0, x1, x2, x3 = v[j]
                v[j] = [x0 * m0[0, 0] + x1 * m0[0, 1], x0 * m0[1, 0] + x1 * m0[1, 1], x2 * m1[0, 0] + x3 * m1[0, 1], x2 * m1[1, 0] + x3 * m1[1, 1]]
        out.reshape(-1, 4)[((n * H + h) * Lt + l) * G:((n * H + h) * Lt + l + 1) * G] = v
    # definition, node by node
    def stream(T, L, xs, w):
        x = T[:, col:col + C].reshape(N, L, H, d) if L else np.zeros((N, 0, H, d), np.float32)
        if norm and L:
            x = x / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + eps) * w
        return x
    x = np.concatenate([stream(Ta, La, xs_a, wa), stream(Tb, Lb, xs_b, wb)], 1)      # [N, Lt, H, d]
    if rope:
        xp = x.reshape(N, Lt, H, d // 2, 2)
        x = np.einsum("lpij,nlhpj->nlhpi", pe.astype(np.float64), xp).reshape(N, Lt, H, d)
    ref = np.transpose(x, (0, 2, 1, 3))                                                   # [N, H, Lt, d] = ggml [d, Lt, H, N]
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("ndv,dkp", [(2, 48), (2, 64), (3, 80), (4, 128), (5, 160)])
def test_flash_row_major_v_tile_transposing_read_fragments(ndv, dkp):
    """k_flash_attn<..., VTR>: V tiles stay row-major [key][dv] in LDS and the PV B-fragments come from ds_read_b64_tr_b16.  Model of the read
    (cdna_hip_programming.md, LDS section): inside each 16-lane group, result element j of lane l is half (l & 3) of the 8 bytes addressed by lane
    4 j + ((l & 15) >> 2).  With the kernel's per-lane addresses every lane must receive, for k-step t and column block nb, the keys
    t*16 + {4 hi .. 4 hi + 3} and the same 8 further on, of output column nb*32 + (lane & 31) — the fragment the transposed-tile path reads —
    and a 32-lane half of one read must touch every LDS bank at most once."""
    dw = ndv * 16
    if dw % 64 not in (16, 48):
        dw += 16
    VRS = dw * 2
    rng = np.random.default_rng(ndv)
    V = rng.standard_normal((64, VRS)).astype(np.float16)
    flat = V.reshape(-1)
    lanes = np.arange(64)
    hi = lanes >> 5
    lane_off = (4 * hi + ((lanes & 15) >> 2)) * VRS + ((lanes >> 4) & 1) * 16 + 4 * (lanes & 3)

    def tr16(addr):  # addr: per-lane element offsets
        out = np.zeros((64, 4), np.float16)
        for l in range(64):
            g16 = l & ~15
            for j in range(4):
                src = g16 + 4 * j + ((l & 15) >> 2)
                out[l, j] = flat[addr[src] + (l & 3)]
        return out

    for t in range(4):
        for nb in range(ndv):
            a = tr16(lane_off + t * 16 * VRS + nb * 32)
            c = tr16(lane_off + (t * 16 + 8) * VRS + nb * 32)
            for l in range(64):
                col = nb * 32 + (l & 31)
                k0 = t * 16 + 4 * (l >> 5)
                np.testing.assert_array_equal(a[l], V[k0:k0 + 4, col])
                np.testing.assert_array_equal(c[l], V[k0 + 8:k0 + 12, col])
            for half in range(2):  # bank = (byte address / 4) % 64, 8 bytes per lane
                banks = []
                for l in range(32 * half, 32 * half + 32):
                    b = ((lane_off[l] + t * 16 * VRS + nb * 32) * 2 // 4) % 64
                    banks += [b, (b + 1) % 64]
                assert len(set(banks)) == 64
    assert dkp <= ndv * 32 and (64 * VRS * 2) % 16 == 0


def _mfma_32x32x16(A, B, C):
    """v_mfma_f32_32x32x16_f16 on per-lane registers: A[l][j] = a[i = l & 31][k = 8 (l >> 5) + j], B[l][j] = b[k = 8 (l >> 5) + j][n = l & 31],
    C[l][r] = c[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31] (the layouts every MFMA kernel of this repo is written against)."""
    a = np.zeros((32, 16), np.float64)
    b = np.zeros((16, 32), np.float64)
    for l in range(64):
        for j in range(8):
            a[l & 31, 8 * (l >> 5) + j] = A[l, j]
            b[8 * (l >> 5) + j, l & 31] = B[l, j]
    d = a @ b
    out = C.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += d[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


@pytest.mark.parametrize("d,Lq,Lk", [(40, 70, 77), (64, 33, 96), (16, 64, 65)])
def test_flash_short_register_resident_kv_indexing(d, Lq, Lk):
    """k_flash_short (attention onto 65..96 keys with the whole K / V in registers): the kernel's staging (K carries the scale), fragment
    addresses, key mask as accumulator init, one-pass softmax with P normalised before packing, and output mapping, replayed lane by lane with
    the MFMA register layouts, must reproduce softmax(Q K^T / sqrt(d)) V."""
    DKP = 48 if d <= 48 else 64
    NDV, KS, KROW, NKB, NKT = 2, DKP // 16, DKP + 8, 3, 6
    VRS = 96  # fa_vtr_stride(2)
    rng = np.random.default_rng(d + Lk)
    q = rng.standard_normal((Lq, d)).astype(np.float32)
    k = rng.standard_normal((Lk, d)).astype(np.float16)
    v = rng.standard_normal((Lk, d)).astype(np.float16)
    scale = 1.0 / np.sqrt(d)
    Ks = np.zeros(96 * KROW, np.float16)
    Vs = np.zeros(96 * VRS, np.float16)
    for e in range(96 * (DKP // 8)):
        key, ch = divmod(e, DKP // 8)
        if key < Lk and ch * 8 < d:  # K carries scale * log2(e)
            Ks[key * KROW + ch * 8: key * KROW + ch * 8 + 8] = (k[key, ch * 8: ch * 8 + 8].astype(np.float32) * np.float32(scale * 1.4426950408889634)).astype(np.float16)
    for e in range(96 * NDV * 4):
        key, ch = divmod(e, NDV * 4)
        if key < Lk and ch * 8 < d:
            Vs[key * VRS + ch * 8: key * VRS + ch * 8 + 8] = v[key, ch * 8: ch * 8 + 8]
    lanes = np.arange(64)
    hi = lanes >> 5
    kf = np.zeros((NKB, KS, 64, 8), np.float16)
    for kb in range(NKB):
        for ks in range(KS):
            for l in range(64):
                o = (kb * 32 + (l & 31)) * KROW + ks * 16 + hi[l] * 8
                kf[kb, ks, l] = Ks[o:o + 8]
    vtr_lane = (4 * hi + ((lanes & 15) >> 2)) * VRS + ((lanes >> 4) & 1) * 16 + 4 * (lanes & 3)

    def tr16(addr):
        out = np.zeros((64, 4), np.float16)
        for l in range(64):
            g16 = l & ~15
            for j in range(4):
                out[l, j] = Vs[addr[g16 + 4 * j + ((l & 15) >> 2)] + (l & 3)]
        return out

    vf = np.zeros((NKT, NDV, 64, 8), np.float16)
    for t in range(NKT):
        for nb in range(NDV):
            p = vtr_lane + t * 16 * VRS + nb * 32
            vf[t, nb, :, :4] = tr16(p)
            vf[t, nb, :, 4:] = tr16(p + 8 * VRS)
    negc = np.zeros((64, 16))  # key mask of the last block = the accumulator's initial value
    for r in range(16):
        negc[64 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Lk, r] = -np.inf
    out = np.full((Lq, d), np.nan, np.float32)
    with np.errstate(invalid="ignore"):
        for q0 in range(0, Lq, 32):
            rowl = np.minimum(lanes & 31, Lq - 1 - q0)
            sc = np.zeros((NKB, 64, 16))
            sc[2] = negc
            for ks in range(KS):
                qf = np.zeros((64, 8), np.float16)
                for l in range(64):
                    d0 = ks * 16 + hi[l] * 8
                    if d0 < d:
                        qf[l] = q[q0 + rowl[l], d0:d0 + 8].astype(np.float16)
                for kb in range(NKB):
                    sc[kb] = _mfma_32x32x16(kf[kb, ks].astype(np.float64), qf.astype(np.float64), sc[kb])
            m = sc[0][:, 0].copy()
            for kb in range(NKB):
                m = np.maximum(m, sc[kb].max(1))
            m = np.maximum(m, m[lanes ^ 32])
            psum = np.zeros(64)
            for kb in range(NKB):
                sc[kb] = np.exp2(sc[kb] - m[:, None])
                psum += sc[kb].sum(1)
            inv = 1.0 / (psum + psum[lanes ^ 32])
            o = np.zeros((NDV, 64, 16))
            for t in range(NKT):
                kb, rb = t >> 1, (t & 1) * 8
                pa = (sc[kb][:, rb:rb + 8] * inv[:, None]).astype(np.float16).astype(np.float64)  # each lane owns ONE query's scores
                for nb in range(NDV):
                    o[nb] = _mfma_32x32x16(pa, vf[t, nb].astype(np.float64), o[nb])
            full = q0 + 32 <= Lq
            for l in range(64):
                for r in range(16):
                    qq = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi[l]
                    if not full and qq >= Lq:
                        continue
                    for nb in range(NDV):
                        dd = nb * 32 + (l & 31)
                        if dd < d:
                            out[qq, dd] = o[nb][l, r]
    s = (q.astype(np.float64) @ k.astype(np.float64).T) * scale
    p = np.exp(s - s.max(1, keepdims=True))
    ref = (p / p.sum(1, keepdims=True)) @ v.astype(np.float64)
    assert np.isfinite(out).all()
    assert np.abs(out - ref).max() < 3e-3 * max(1.0, np.abs(ref).max())


# ---- round 4 ------------------------------------------------------------------------------------------------------------------------
def permlane16_swap(vdst: np.ndarray, src: np.ndarray):
    """v_permlane16_swap_b32 on 64-lane registers: lanes 16..31 of vdst <-> lanes 0..15 of src, lanes 48..63 of vdst <-> lanes 32..47 of src"""
    d, s = vdst.copy(), src.copy()
    for base in (0, 32):
        d[base + 16:base + 32], s[base:base + 16] = src[base:base + 16].copy(), vdst[base + 16:base + 32].copy()
    return d, s


@pytest.mark.parametrize("inner,cb_per_wave", [(160, 5), (64, 2), (80, 5)])
def test_geglu16_interleaved_image_and_lane_pairing(inner, cb_per_wave):
    """epi_geglu16 (g16_common.h) + k_wswz_linear(geglu_inner < 0) (wgemm.hip): the FF1 weight image interleaves value and gate at 16 columns, a 32x32
    accumulator block then holds value(o) in lane l and gate(o) in lane l ^ 16 of the same register, and ONE permlane16_swap per register pair (j, j + 8)
    brings them together: every (row, output) of the block must be finished exactly once, by the lane the store address is computed for, from
    the matching value / gate pair."""
    rng = np.random.default_rng(inner)
    R = 2 * inner
    nblk = (R + 31) // 32
    # image column (= weight row of the image) -> source row: block b, lane-row w: out = 16 b + (w & 15); source = out (w < 16) or inner + out; beyond inner: zero row
    src_row = np.full(nblk * 32, -1)
    for rb in range(nblk):
        for w in range(32):
            out = rb * 16 + (w & 15)
            if out < inner:
                src_row[rb * 32 + w] = (inner if (w & 16) else 0) + out
    assert sorted(r for r in src_row if r >= 0) == list(range(R))           # every value and gate row appears exactly once
    # a "GEMM" whose output column c is the image column c: y[row][c] = f(row, src_row[c]) — use distinct numbers per (row, source row)
    rows = 32
    full = rng.standard_normal((rows, R)).astype(np.float32)               # full[row][source row]: value columns [0, inner), gate columns [inner, 2 inner)
    done = np.zeros((rows, inner), dtype=int)
    got_v = np.zeros((rows, inner), np.float32)
    got_g = np.zeros((rows, inner), np.float32)
    for cb in range(cb_per_wave):                                           # the column blocks of one wave (col0 = 0, wc = 0)
        oc0 = cb * 16
        if oc0 >= inner:
            continue
        # accumulator block in the D[row][col] layout: register r of lane (hi, lc) holds row (r & 3) + 8 (r >> 2) + 4 hi, column 32 cb + lc
        acc = np.zeros((16, 64), np.float32)
        for lane in range(64):
            hi, lc = lane >> 5, lane & 31
            sr = src_row[cb * 32 + lc]
            for r in range(16):
                row = (r & 3) + 8 * (r >> 2) + 4 * hi
                acc[r, lane] = full[row, sr] if sr >= 0 else 0.0
        for j in range(8):
            v, g = permlane16_swap(acc[j + 8], acc[j])                      # sw[0], sw[1] of the kernel
            for lane in range(64):
                hi, l16, up = lane >> 5, lane & 15, (lane >> 4) & 1
                r = j if up else j + 8
                row = (r & 3) + 8 * (r >> 2) + 4 * hi
                o = oc0 + l16
                done[row, o] += 1
                got_v[row, o], got_g[row, o] = v[lane], g[lane]
    covered = min(inner, cb_per_wave * 16)
    assert (done[:, :covered] == 1).all() and (done[:, covered:] == 0).all()
    np.testing.assert_array_equal(got_v[:, :covered], full[:, :covered])
    np.testing.assert_array_equal(got_g[:, :covered], full[:, inner:inner + covered])


@pytest.mark.parametrize("C1,C2,hw,groups", [(64, 32, 16, 32), (640, 320, 64, 32), (320, 320, 256, 32), (128, 64, 4, 32)])
def test_group_norm_two_source_float4_indexing(C1, C2, hw, groups):
    """gn_ld4 (gemm16.hip): float4 i4 of the (image n, channels c0..) slab of the NEVER-BUILT concatenation [x (C1 channels) | x2 (C2)] — hw % 4 == 0, so the
    four elements share their channel and their source; groups may straddle the two sources (C1 not a multiple of the group width)."""
    rng = np.random.default_rng(C1 + hw)
    N, C = 2, C1 + C2
    a = rng.standard_normal((N, C1, hw)).astype(np.float32)
    b = rng.standard_normal((N, C2, hw)).astype(np.float32)
    cat = np.concatenate([a, b], axis=1)
    cpg = (C + groups - 1) // groups

    def gn_ld4(n, c0, i4):
        e = i4 * 4
        ch = c0 + e // hw
        of = e - (ch - c0) * hw
        src = a[n, ch] if ch < C1 else b[n, ch - C1]
        return src[of:of + 4]

    for n in range(N):
        for gidx in range(groups):
            c0, c1 = gidx * cpg, min(gidx * cpg + cpg, C)
            if c0 >= c1:
                continue
            flat = cat[n, c0:c1].reshape(-1)
            for i4 in rng.integers(0, flat.size // 4, 16):
                np.testing.assert_array_equal(gn_ld4(n, c0, int(i4)), flat[4 * i4:4 * i4 + 4])


@pytest.mark.parametrize("T,NT,G,hybrid", [(208, 64, 256, False), (204, 96, 256, False), (576, 96, 256, False), (408, 96, 256, True), (612, 64, 256, True),
                                           (1428, 96, 256, True), (5, 7, 8, False), (19, 5, 8, True)])
def test_stream_k_unit_ranges_parts_and_slab_slots(T, NT, G, hybrid):
    """k_gemm16<..., SK> (gemm16.hip): the (tile, K-tile) units of the cut part are split into G equal contiguous ranges; a tile cut by a range boundary is a
    list of parts in K order, part p computed by (logical) workgroup wf + p with wf = owner of the tile's first unit = floor(((u + 1) G - 1) / U); a workgroup
    dumps a part with kt0 > 0 into slab slot 2 w and a part starting at kt0 == 0 into slot 2 w + 1.  Checked here: every unit is covered exactly once, the owner
    formula agrees with the ranges, the slots a reducer reads are the slots the parts were written to and no slot is used twice; hybrid: full rounds stay
    whole tiles, G per round, contiguous per workgroup."""
    dp = (T // G) * G if hybrid else 0
    Tr = T - dp
    U = Tr * NT
    cover = np.zeros((T, NT), dtype=int)
    slot_of = {}      # (tile, part index) -> slot written
    used = set()
    for w in range(G):
        u, end = w * U // G, (w + 1) * U // G
        partials = 0
        while u < end:
            rt = u // NT
            kt0 = u - rt * NT
            nt = min(NT - kt0, end - u)
            cover[dp + rt, kt0:kt0 + nt] += 1
            if not (kt0 == 0 and nt == NT):
                uf = rt * NT
                wf, wl = ((uf + 1) * G - 1) // U, ((uf + NT) * G - 1) // U
                assert wf <= w <= wl
                slot = 2 * w + (1 if kt0 == 0 else 0)
                assert slot not in used
                used.add(slot)
                slot_of[(rt, w - wf)] = slot
                partials += 1
            u += nt
        assert partials <= 2
        per = dp // G
        for t in range(w * per, w * per + per):
            cover[t, :] += 1
    assert (cover == 1).all()
    # the reducer of tile rt reads part p from slot (p == 0 ? 2 wf + 1 : 2 (wf + p))
    for rt in range(Tr):
        uf = rt * NT
        if U == 0:
            break
        wf, wl = ((uf + 1) * G - 1) // U, ((uf + NT) * G - 1) // U
        if wl == wf:
            assert (rt, 0) not in slot_of     # a whole tile: no slab traffic
            continue
        for p in range(wl - wf + 1):
            assert slot_of[(rt, p)] == (2 * wf + 1 if p == 0 else 2 * (wf + p))


# ---- round 5: row-split launches (gemm16.hip g16_tail_rows) and the clamped weight fetch of half-empty 256-column tiles (wblk_lim) -------------------
def tail_rows(rows, M, cus=256):
    """restatement of g16_tail_rows: row tiles (256 rows) of the main launch, 0 = one launch"""
    ncol, rt = (M + 255) // 256, (rows + 255) // 256
    T = rt * ncol
    full, rem = T // cus, T % cus
    if full < 1 or rem == 0:
        return 0
    rtm = full * cus // ncol
    if rtm <= 0 or rtm >= rt:
        return 0
    tail = rows - rtm * 256
    wgs = ((tail + 127) // 128) * ((M + 127) // 128)
    tail_cost = 0.5 if wgs <= 256 else (0.75 if wgs <= 512 else 1.25 * ((wgs + 767) // 768))
    main_cost = (rtm * ncol + cus - 1) // cus
    return rtm if main_cost + tail_cost < (full + 1) * 0.95 else 0


@pytest.mark.parametrize("rows,M,expect_split", [(4352, 12288, True), (4352, 9216, True), (4352, 21504, False), (4352, 3072, False), (8192, 2432, True),
                                                 (8500, 9728, True), (4096, 12288, False), (65536, 320, False), (300, 4096, False)])
def test_row_split_covers_every_row_exactly_once(rows, M, expect_split):
    rtm = tail_rows(rows, M)
    assert (rtm > 0) == expect_split, (rows, M, rtm)
    if not rtm:
        return
    ncol = (M + 255) // 256
    # main launch: row tiles [0, rtm) x every column tile, whole rounds at most; tail launch: row_base = rtm * 256, rows - row_base rows on ITS tile size
    assert rtm * ncol <= (((rows + 255) // 256) * ncol // 256) * 256
    row_base = rtm * 256
    covered = np.zeros(rows, dtype=np.int32)
    for t in range(rtm):
        covered[t * 256:min(rows, (t + 1) * 256)] += 1
    for bm in (128, 256):   # whatever tile the tail picks: row0 = row_base + tile * BM, rows >= R are masked by the epilogue
        c2 = covered.copy()
        for t in range((rows - row_base + bm - 1) // bm):
            r0 = row_base + t * bm
            c2[r0:min(rows, r0 + bm)] += 1
        assert (c2 == 1).all()
    # a nested split of the tail would start at row_base again: the recursion passes the remaining rows and the accumulated base
    assert row_base % 256 == 0 and 0 < rows - row_base < rows


@pytest.mark.parametrize("M", [2432, 7296, 2176, 3200])
def test_half_empty_column_tile_fetches_stay_inside_the_weight_image(M):
    """the weight image holds rup128(M) columns = rup128(M) / 32 fragment blocks; a 256-column tile asks for blocks col0/32 .. col0/32 + 7: the blocks at or
    beyond the limit are clamped to the last block of the image (their outputs are masked by col < M)"""
    lim = ((M + 127) // 128 * 128) // 32
    ncol = (M + 255) // 256
    for ct in range(ncol):
        for cb in range(8):
            wb = ct * 8 + cb
            wbc = min(wb, lim - 1) if M % 256 else wb
            assert 0 <= wbc < lim
            if wb * 32 < M:
                assert wbc == wb   # every block that holds real columns is fetched from its own place


# ---- round 5: weight-major workgroup order (g16_common.h g16_wg_order) -----------------------------------------------------------------------------------
def wg_order(x, y, gx, ncol, worder):
    """restatement of g16_wg_order: workgroup (x, y) of a gx x S grid -> (tile id, K slice)"""
    if worder > 0:
        L = y * gx + x
        k, j = L & 7, L >> 3
        pl, r = divmod(j, worder)
        p = pl * 8 + k
        ks, ct = divmod(p, ncol)
        return r * ncol + ct, ks
    bid = (x & 7) * (gx >> 3) + (x >> 3) if gx % 8 == 0 else x
    return bid, y


@pytest.mark.parametrize("nrow,ncol,S", [(4, 4, 16), (16, 4, 4), (16, 4, 8), (4, 8, 2), (2, 4, 2), (8, 8, 1), (16, 8, 3)])
def test_weight_major_order_is_a_bijection_and_keeps_a_weight_chunk_on_one_xcd(nrow, ncol, S):
    """8x8 / 16x16-level convs (4 x 4 tiles x 16 slices; 16 x 4 tiles x 4 slices): every (tile, slice) is computed exactly once, every (column tile, slice)
    weight chunk is touched by ONE XCD only (workgroup L = y * gx + x runs on XCD L % 8), and its row tiles are consecutive in that XCD's dispatch order."""
    gx = nrow * ncol
    assert gx % 8 == 0 and (ncol * S) % 8 == 0   # the launcher's conditions (gemm16_worder_rows)
    seen, xcd_of, order = set(), {}, {}
    for y in range(S):
        for x in range(gx):
            bid, ks = wg_order(x, y, gx, ncol, nrow)
            assert 0 <= bid < gx and 0 <= ks < S
            assert (bid, ks) not in seen
            seen.add((bid, ks))
            L = y * gx + x
            chunk = (bid % ncol, ks)
            assert xcd_of.setdefault(chunk, L % 8) == L % 8
            order.setdefault(chunk, []).append(L >> 3)
    assert len(seen) == gx * S
    for chunk, js in order.items():
        js = sorted(js)
        assert js == list(range(js[0], js[0] + nrow))   # back to back on that XCD
    # the default order, for contrast: a weight chunk is fetched by several XCDs
    xcds = {}
    for y in range(S):
        for x in range(gx):
            bid, ks = wg_order(x, y, gx, ncol, 0)
            xcds.setdefault((bid % ncol, ks), set()).add((y * gx + x) % 8)
    assert max(len(v) for v in xcds.values()) > 1 or nrow * ncol <= 8
