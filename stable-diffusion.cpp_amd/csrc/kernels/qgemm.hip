// qgemm.hip — Linear layers whose weights are q8_0 / q4_0 GGUF blocks and whose activation has only one or two rows (adaLN / modulation
// vectors of the DiTs and the ResBlock embedding projections: 1 row per image).  Such a
// contraction is a pure WEIGHT STREAM: the kernel reads the RAW quantised blocks from HBM exactly once (34 B / 18 B per 32 weights,
// coalesced 16-byte loads of whole row segments), dequantises in registers and never builds the f16 weight image the MFMA GEMMs use
// (that image is 3.5x the bytes of q4_0).
//
// Arithmetic = the MFMA path's rounding points, so that a Linear gives the same values (to f32 summation order and the f16 rounding of
// the dequantised weight, ~1e-4) whichever kernel its row count selects — a batch of 1 and a batch of 8 must agree: activations are
// rounded to f16, weights are d * q exactly, products and sums are f32.  (ggml-cpu quantises the activations to q8_0 instead, SURVEY.md
// Appendix E.1; the first version of this kernel did the same on v_dot4_i32_i8 and moved the full-width SDXL forward from 1.5e-3 to 7e-3
// of the exact-weight oracle — r02, profiles/r02e_fullwidth_parity.txt vs the failing run — so it was replaced.)
//   q8_0:  sum_i q_i x_i           = sum_i u_i x_i - 128 * sum_i x_i     with u = q ^ 0x80 (unsigned bytes -> v_cvt_f32_ubyteN)
//   q4_0:  sum_i (n_i - 8) x_i     = sum_i n_i x_i -   8 * sum_i x_i     with the nibbles n spread to bytes by two AND / shift pairs
//
// Mapping: a wave owns CPW consecutive weight rows (= output features) and walks K in segments of 64 blocks — lane l of the wave owns
// block l of the segment for every row, so the activation values of the segment sit in registers (R rows x 32 floats) and are reused
// by all CPW weight rows.  A weight row segment (64 x 34 B = 2176 B, or 64 x 18 B = 1152 B) is fetched with 16-byte loads of the whole
// contiguous range into a per-wave LDS strip and re-read block-wise with aligned 4-byte LDS loads + v_alignbit (the blocks are only
// 2-byte aligned).  Cross-lane sums at the end, bias / residual in the store.
#include "device_utils.h"
#include "kernels.h"
#include "ktime.h"

namespace mi355x {

struct QGArgs {
    const char* W;       // raw quantised rows
    int64_t row_bytes;
    const float* x;      // activation rows, f32 (rounded to f16 in the kernel)
    int64_t xs;          // row stride in floats
    float* dst;
    int64_t ldd;
    const float* bias;
    const float* residual;  // same layout as dst
    float scale, pre_scale;
    int K, M, rows;
};

// QT: 8 = q8_0 (34-byte blocks), 4 = q4_0 (18-byte blocks).  R = activation rows held in registers (rows <= R), CPW = weight rows per wave.
template <int QT, int R, int CPW>
__global__ __launch_bounds__(256) void k_qgemv(QGArgs g) {
    constexpr int BLK  = QT == 8 ? 34 : 18;
    constexpr int SEGB = 64 * BLK;               // bytes of one row segment (2176 / 1152)
    constexpr int NG   = (SEGB + 15) / 16;       // 16-byte granules per segment (136 / 72)
    constexpr int NLD  = (NG + 63) / 64;         // loads per lane per segment (3 / 2)
    constexpr int NDW  = QT == 8 ? 9 : 5;        // aligned dwords covering one block at any 2-byte phase
    __shared__ __attribute__((aligned(16))) char strip[4][NLD * 64 * 16 + 16];
    // activation rows as f16, staged ONCE per workgroup with coalesced loads (16-byte chunk c of block b sits at chunk slot c ^ (b & 3): a
    // lane's four chunk reads then spread over the banks).  Every wave re-reading its blocks straight from global memory — 8 loads of one
    // 128-byte line per lane — was 7x the address-path work of the weight stream itself.
    extern __shared__ __attribute__((aligned(16))) char xlds[];  // [R][K] halfs
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* my       = strip[wave];
    const int nblk = g.K / 32;
    const int nseg = (nblk + 63) / 64;
    const int col0 = (blockIdx.x * 4 + wave) * CPW;
    for (int t = 0; t < R; ++t) {
        const int tt    = t < g.rows ? t : 0;
        const float* xr = g.x + (int64_t)tt * g.xs;
        for (int c8 = threadIdx.x; c8 < g.K / 8; c8 += 256) {  // 8 values = one 16-byte f16 chunk
            const float4 a = *(const float4*)(xr + c8 * 8), b = *(const float4*)(xr + c8 * 8 + 4);
            half8_t h;
            h[0] = (_Float16)(a.x * g.pre_scale); h[1] = (_Float16)(a.y * g.pre_scale); h[2] = (_Float16)(a.z * g.pre_scale); h[3] = (_Float16)(a.w * g.pre_scale);
            h[4] = (_Float16)(b.x * g.pre_scale); h[5] = (_Float16)(b.y * g.pre_scale); h[6] = (_Float16)(b.z * g.pre_scale); h[7] = (_Float16)(b.w * g.pre_scale);
            const int blk = c8 >> 2, ch = c8 & 3;
            *(half8_t*)(xlds + ((size_t)t * g.K + (size_t)blk * 32) * 2 + ((ch ^ (blk & 3)) << 4)) = h;
        }
    }
    __syncthreads();
    if (col0 >= g.M) return;

    float acc[CPW][R];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int t = 0; t < R; ++t) acc[c][t] = 0.f;

    const uint32_t boff = (uint32_t)lane * BLK;        // byte offset of this lane's block inside the strip (even)
    const uint32_t bal  = boff & ~3u;                  // aligned-down dword address
    const uint32_t bsh  = (boff & 2u) * 8u;            // 0 or 16: bit offset of the block inside the first dword

    for (int seg = 0; seg < nseg; ++seg) {
        const int blk      = seg * 64 + lane;
        const bool have    = blk < nblk;
        const int seg_blks = min(64, nblk - seg * 64);
        const int seg_ng   = (seg_blks * BLK + 15) / 16;  // granules that hold real bytes of this row
        // activation values of this lane's block: registers, shared by all CPW weight rows; rounded to f16 like every MFMA operand
        float xf[R][32], xsum[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int tt = t < g.rows ? t : 0;
            xsum[t]      = 0.f;
            if (have && t < g.rows) {
                const char* xb = xlds + ((size_t)tt * g.K + (size_t)blk * 32) * 2;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const half8_t h = *(const half8_t*)(xb + ((ch ^ (blk & 3)) << 4));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xf[t][8 * ch + j] = (float)h[j];
                        xsum[t] += xf[t][8 * ch + j];
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) xf[t][j] = 0.f;
            }
        }
        const int64_t seg_byte = (int64_t)seg * SEGB;
        // ---- coalesced 16-byte loads of the contiguous segment of ALL CPW rows first (independent loads: CPW x 1-2 KB in flight per wave —
        // issued one row at a time behind the LDS round trip of the previous row they serialised on the full HBM latency: 740 GB/s)
        uint4 gl[CPW][NLD];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int col    = min(col0 + c, g.M - 1);
            const char* rowp = g.W + (int64_t)col * g.row_bytes + seg_byte;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int gidx = i * 64 + lane;
                gl[c][i]       = gidx < seg_ng ? *(const uint4*)(rowp + (int64_t)gidx * 16) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int col = col0 + c;
            if (col >= g.M) break;  // wave-uniform
            // the strip is private to this wave and LDS operations of one wave execute in order: a wavefront-scope fence (no instruction, it
            // only stops the compiler from moving the block reads above other lanes' stores) is all the synchronisation needed
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < NLD; ++i) *(uint4*)(my + (i * 64 + lane) * 16) = gl[c][i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- this lane's block: NDW aligned dwords, funnel-shifted to the block's 2-byte phase
            uint32_t raw[NDW];
#pragma unroll
            for (int j = 0; j < NDW; ++j) raw[j] = *(const uint32_t*)(my + bal + 4 * j);
            uint32_t wq[NDW - 1];
#pragma unroll
            for (int j = 0; j + 1 < NDW; ++j) wq[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], bsh);  // bytes [4j + phase, 4j + phase + 4)
            // wq[0] = {d (f16), first two quant bytes}; quant dword m = bytes 2 + 4m .. : alignbit(wq[m + 1], wq[m], 16)
            const float dw = (float)__builtin_bit_cast(_Float16, (uint16_t)(wq[0] & 0xFFFFu));
            constexpr int NQ = QT == 8 ? 8 : 4;
            uint32_t qd[NQ];
#pragma unroll
            for (int m = 0; m < NQ; ++m) {
                const uint32_t hi = m + 1 < NDW - 1 ? wq[m + 1] : (raw[NDW - 1] >> bsh);
                qd[m]             = __builtin_amdgcn_alignbit(hi, wq[m], 16);
            }
            if (have) {
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    float sa = 0.f, sb = 0.f;  // two chains: the 32 FMAs of a block are dependent otherwise
                    if (QT == 8) {
#pragma unroll
                        for (int m = 0; m < 8; ++m) {
                            const uint32_t u = qd[m] ^ 0x80808080u;
                            sa = fmaf((float)(u & 0xFFu), xf[t][4 * m], sa);
                            sb = fmaf((float)((u >> 8) & 0xFFu), xf[t][4 * m + 1], sb);
                            sa = fmaf((float)((u >> 16) & 0xFFu), xf[t][4 * m + 2], sa);
                            sb = fmaf((float)(u >> 24), xf[t][4 * m + 3], sb);
                        }
                        acc[c][t] += dw * ((sa + sb) - 128.f * xsum[t]);
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const uint32_t lo = qd[m] & 0x0F0F0F0Fu, hi = (qd[m] >> 4) & 0x0F0F0F0Fu;  // elements 4m.. and 16 + 4m..
                            sa = fmaf((float)(lo & 0xFFu), xf[t][4 * m], sa);
                            sb = fmaf((float)((lo >> 8) & 0xFFu), xf[t][4 * m + 1], sb);
                            sa = fmaf((float)((lo >> 16) & 0xFFu), xf[t][4 * m + 2], sa);
                            sb = fmaf((float)(lo >> 24), xf[t][4 * m + 3], sb);
                            sa = fmaf((float)(hi & 0xFFu), xf[t][16 + 4 * m], sa);
                            sb = fmaf((float)((hi >> 8) & 0xFFu), xf[t][16 + 4 * m + 1], sb);
                            sa = fmaf((float)((hi >> 16) & 0xFFu), xf[t][16 + 4 * m + 2], sa);
                            sb = fmaf((float)(hi >> 24), xf[t][16 + 4 * m + 3], sb);
                        }
                        acc[c][t] += dw * ((sa + sb) - 8.f * xsum[t]);
                    }
                }
            }
        }
    }
    // ---- cross-lane sums and store
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int col = col0 + c;
        if (col >= g.M) break;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const float v = wave_sum(acc[c][t]);
            if (lane == 0 && t < g.rows) {
                float o = v * g.scale + (g.bias ? g.bias[col] : 0.f);
                if (g.residual) o += g.residual[(int64_t)t * g.ldd + col];
                g.dst[(int64_t)t * g.ldd + col] = o;
            }
        }
    }
}

size_t qgemv_workspace_bytes(int64_t, int64_t) { return 0; }  // (the first version quantised the activations into a workspace)

bool qgemv_supported(int wtype, int64_t rows, int64_t K) {
    // whole row segments are fetched with 16-byte loads: every row must start 16-byte aligned (34 * K/32 and 18 * K/32 are multiples of 16 iff
    // K % 256 == 0).  Rows: the activation values of a block live in registers (32 per row) — one or two rows.  Above that the f16 weight
    // image + MFMA GEMM take over: from ~32 rows on the contraction stops being a pure weight stream anyway.
    return (wtype == 8 || wtype == 2) && rows >= 1 && rows <= 2 && K % 256 == 0 && K >= 256 && K <= 12288;  // activations staged as f16 in LDS: 2 rows x 12288 halfs = 48 KB
}

// x: f32 rows (row stride xs floats, 16-byte aligned), multiplied by pre_scale before the f16 rounding (ggml_ext_linear's scale)
void launch_qgemv(hipStream_t s, float* dst, int64_t ldd, const float* x, int64_t xs, int64_t rows, const void* wraw, int wtype, int64_t K, int64_t M, void*,
                  const Epilogue& ep, float pre_scale) {
    const int64_t nblk  = K / 32;
    const size_t wbytes = (size_t)M * (size_t)nblk * (wtype == 8 ? 34 : 18);
    KScope ks_(s, KF_QGEMM, 2.0 * rows * K * M, (double)wbytes + (double)rows * K * 4.0 + (double)rows * M * 4.0);
    QGArgs g;
    g.W         = (const char*)wraw;
    g.row_bytes = nblk * (wtype == 8 ? 34 : 18);
    g.x = x; g.xs = xs;
    g.dst = dst; g.ldd = ldd;
    g.bias = ep.bias; g.residual = ep.residual; g.scale = ep.scale; g.pre_scale = pre_scale;
    g.K = (int)K; g.M = (int)M; g.rows = (int)rows;
    constexpr int CPW = 4;
    const unsigned grid = (unsigned)((M + 4 * CPW - 1) / (4 * CPW));
#define QG_LAUNCH(QT_, R_) k_qgemv<QT_, R_, CPW><<<grid, 256, (size_t)(R_) * K * 2, s>>>(g)
    if (wtype == 8) {
        if (rows == 1) QG_LAUNCH(8, 1);
        else QG_LAUNCH(8, 2);
    } else {
        if (rows == 1) QG_LAUNCH(4, 1);
        else QG_LAUNCH(4, 2);
    }
#undef QG_LAUNCH
}

}  // namespace mi355x
