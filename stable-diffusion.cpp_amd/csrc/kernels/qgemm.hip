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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "device_utils.h"
#include "kernels.h"
#include "ktime.h"

namespace mi355x {

// Sum NV per-lane partial values over the 64 lanes of a wave with NV - 1 + (6 - log2 NV) cross-lane exchanges instead of 6 NV: every halving
// step sends the half of the values the partner lane keeps (lanes with bit o set keep the upper half).  Afterwards value index
// (lane >> (6 - log2 NV)) is complete in every lane of its group.  (k_fgemv's epilogue did 32 butterfly sums = 192 dependent ds_bpermute
// round trips: a third of its 32 us, profiles/r02u_kernel_stats.csv.)
template <int NV>
__device__ __forceinline__ float wave_sum_transpose(float (&v)[NV], int lane) {
    static_assert(NV >= 1 && NV <= 32 && (NV & (NV - 1)) == 0, "NV must be a power of two <= 32");
    constexpr int LOG = NV == 1 ? 0 : NV == 2 ? 1 : NV == 4 ? 2 : NV == 8 ? 3 : NV == 16 ? 4 : 5;
#pragma unroll
    for (int st = 0; st < LOG; ++st) {
        const int n = NV >> st, o = 32 >> st;
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float send = up ? v[i] : v[i + n / 2];
            const float keep = up ? v[i + n / 2] : v[i];
            v[i]             = keep + __shfl_xor(send, o, 64);
        }
    }
    float r = v[0];
#pragma unroll
    for (int o = 32 >> LOG; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
    return r;
}

// grouped launch (launch_qgemv_group): the M output columns are the concatenation of several Linears that read the SAME activation rows — output column
// `start` .. of the concatenation is row 0 .. of member weight W, with its own bias.  No concatenated copy of the weights exists: the wave finds its member.
struct QGMember {
    const char* W;
    const float* bias;
    int start, pad_;
};
struct QGArgs {
    const QGMember* members = nullptr;  // device table, sorted by start; nullptr: one weight (W / bias below)
    int n_members           = 0;
    const char* W;       // raw quantised rows
    int64_t row_bytes;
    const float* x;      // activation rows, f32 (rounded to f16 in the kernel)
    int64_t xs;          // row stride in floats
    float* dst;
    int64_t ldd;
    const float* bias;
    const float* residual;  // same layout as dst
    float scale, pre_scale;
    int K, M, rows, pre_silu;
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
            float4 a = *(const float4*)(xr + c8 * 8), b = *(const float4*)(xr + c8 * 8 + 4);
            if (g.pre_silu) {  // the SiLU node in front of this Linear was deferred to here (planner: presilu)
                a.x = act_apply<UN_SILU>(a.x); a.y = act_apply<UN_SILU>(a.y); a.z = act_apply<UN_SILU>(a.z); a.w = act_apply<UN_SILU>(a.w);
                b.x = act_apply<UN_SILU>(b.x); b.y = act_apply<UN_SILU>(b.y); b.z = act_apply<UN_SILU>(b.z); b.w = act_apply<UN_SILU>(b.w);
            }
            half8_t h;
            h[0] = (_Float16)(a.x * g.pre_scale); h[1] = (_Float16)(a.y * g.pre_scale); h[2] = (_Float16)(a.z * g.pre_scale); h[3] = (_Float16)(a.w * g.pre_scale);
            h[4] = (_Float16)(b.x * g.pre_scale); h[5] = (_Float16)(b.y * g.pre_scale); h[6] = (_Float16)(b.z * g.pre_scale); h[7] = (_Float16)(b.w * g.pre_scale);
            const int blk = c8 >> 2, ch = c8 & 3;
            *(half8_t*)(xlds + ((size_t)t * g.K + (size_t)blk * 32) * 2 + ((ch ^ (blk & 3)) << 4)) = h;
        }
    }
    __syncthreads();
    if (col0 >= g.M) return;
    // this wave's weight rows: of the single weight, or of the member of a grouped launch that owns output column col0 (member sizes are multiples of
    // CPW, so the wave's CPW columns belong to one member); Wb / bb are biased so that they are indexed by the GLOBAL column like g.W / g.bias
    const char* Wb  = g.W;
    const float* bb = g.bias;
    if (g.members) {
        int m = 0;
        while (m + 1 < g.n_members && g.members[m + 1].start <= col0) ++m;
        const QGMember mem = g.members[m];
        Wb = mem.W - (int64_t)mem.start * g.row_bytes;
        bb = mem.bias ? mem.bias - mem.start : nullptr;
    }

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
            const int col    = min(col0 + c, g.M - 1);  // (grouped: the clamp only bites in the last member)
            const char* rowp = Wb + (int64_t)col * g.row_bytes + seg_byte;
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
                float o = v * g.scale + (bb ? bb[col] : 0.f);
                if (g.residual) o += g.residual[(int64_t)t * g.ldd + col];
                g.dst[(int64_t)t * g.ldd + col] = o;
            }
        }
    }
}

// k_qgemv_rows — the same raw-block stream for 3 .. 16 activation rows (modulation vectors / embedders of a DiT batch, SDXL label embedding at
// batch > 1).  A block's 32 weights are decoded ONCE into registers (exact integers, the scale d applied to the block sum) and re-used by every
// row; the rows live in LDS as f16 (R x 32 floats per lane do not fit in registers beyond two rows).  Rounding points as k_qgemv.  One launch:
// no f16 pack of the rows, no weight image, no split-K pass (r02r: the MFMA paths need ~30 us for such a Linear, most of it launch chain).
template <int QT, int R, int CPW>
__global__ __launch_bounds__(256) void k_qgemv_rows(QGArgs g) {
    constexpr int BLK  = QT == 8 ? 34 : 18;
    constexpr int SEGB = 64 * BLK;
    constexpr int NG   = (SEGB + 15) / 16;
    constexpr int NLD  = (NG + 63) / 64;
    constexpr int NDW  = QT == 8 ? 9 : 5;
    __shared__ __attribute__((aligned(16))) char strip[4][NLD * 64 * 16 + 16];
    extern __shared__ __attribute__((aligned(16))) char xlds[];  // [R][K] halfs, 16-byte chunk c of block b at chunk slot c ^ (b & 3)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* my       = strip[wave];
    const int nblk = g.K / 32;
    const int nseg = (nblk + 63) / 64;
    const int col0 = (blockIdx.x * 4 + wave) * CPW;
    for (int c8 = threadIdx.x; c8 < g.K / 8; c8 += 256) {  // all R rows of a chunk column in flight together
        float4 a[R], b[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const float* xr = g.x + (int64_t)(t < g.rows ? t : 0) * g.xs + c8 * 8;
            a[t]            = *(const float4*)xr;
            b[t]            = *(const float4*)(xr + 4);
        }
        const int blk = c8 >> 2, ch = c8 & 3;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            float v[8] = {a[t].x, a[t].y, a[t].z, a[t].w, b[t].x, b[t].y, b[t].z, b[t].w};
            half8_t h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = g.pre_silu ? act_apply<UN_SILU>(v[j]) : v[j];
                h[j]          = (_Float16)(u * g.pre_scale);
            }
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
    const uint32_t boff = (uint32_t)lane * BLK;
    const uint32_t bal  = boff & ~3u;
    const uint32_t bsh  = (boff & 2u) * 8u;
    for (int seg = 0; seg < nseg; ++seg) {
        const int blk      = seg * 64 + lane;
        const bool have    = blk < nblk;
        const int seg_blks = min(64, nblk - seg * 64);
        const int seg_ng   = (seg_blks * BLK + 15) / 16;
        const int64_t seg_byte = (int64_t)seg * SEGB;
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
        float wd[CPW][32];  // exact integers q (q8_0) / n - 8 (q4_0)
        float dw[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < NLD; ++i) *(uint4*)(my + (i * 64 + lane) * 16) = gl[c][i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t raw[NDW];
#pragma unroll
            for (int j = 0; j < NDW; ++j) raw[j] = *(const uint32_t*)(my + bal + 4 * j);
            uint32_t wq[NDW - 1];
#pragma unroll
            for (int j = 0; j + 1 < NDW; ++j) wq[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], bsh);
            dw[c] = have ? (float)__builtin_bit_cast(_Float16, (uint16_t)(wq[0] & 0xFFFFu)) : 0.f;
            constexpr int NQ = QT == 8 ? 8 : 4;
#pragma unroll
            for (int m = 0; m < NQ; ++m) {
                const uint32_t hi = m + 1 < NDW - 1 ? wq[m + 1] : (raw[NDW - 1] >> bsh);
                const uint32_t q  = __builtin_amdgcn_alignbit(hi, wq[m], 16);
                if (QT == 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wd[c][4 * m + j] = (float)(int)(int8_t)(q >> (8 * j));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        wd[c][4 * m + j]      = (float)(int)((q >> (8 * j)) & 0xFu) - 8.f;
                        wd[c][16 + 4 * m + j] = (float)(int)((q >> (8 * j + 4)) & 0xFu) - 8.f;
                    }
                }
            }
        }
        const int bx = have ? blk : 0;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const char* xb = xlds + ((size_t)t * g.K + (size_t)bx * 32) * 2;
            float xv[32];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const half8_t h = *(const half8_t*)(xb + ((ch ^ (bx & 3)) << 4));
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[8 * ch + j] = (float)h[j];
            }
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int j = 0; j < 32; j += 2) {
                    sa = fmaf(wd[c][j], xv[j], sa);
                    sb = fmaf(wd[c][j + 1], xv[j + 1], sb);
                }
                acc[c][t] += dw[c] * (sa + sb);  // dw = 0 for lanes past the last block
            }
        }
    }
    {
        constexpr int NV  = CPW * R;  // value index c * R + t
        constexpr int GRP = 64 / NV;  // lanes per finished value
        float v[NV];
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int t = 0; t < R; ++t) v[c * R + t] = acc[c][t];
        const float tot = wave_sum_transpose<NV>(v, lane);
        const int idx = lane / GRP, c = idx / R, t = idx - c * R, col = col0 + c;
        if ((lane & (GRP - 1)) == 0 && t < g.rows && col < g.M) {
            float o = tot * g.scale + (g.bias ? g.bias[col] : 0.f);
            if (g.residual) o += g.residual[(int64_t)t * g.ldd + col];
            g.dst[(int64_t)t * g.ldd + col] = o;
        }
    }
}

// option "qgemv_max_rows" (<= 16).  Default 4: k_qgemv_rows is VALU work proportional to the row count and holds R x K halfs of LDS per workgroup;
// on the DiT modulation / text-stream shapes it beats the f16-image GEMM path up to 4 rows (3072 -> 9216 q8_0: 24 vs 31 us, 4096 -> 3072: 14 vs 29 us)
// and loses from 8 rows on (36 vs 28 us, 16 rows 89 vs 33 us) — profiles/r02s_qgemm_paths_probe.txt
static int g_qgemv_max_rows = 4;
void qgemv_set_max_rows(int v) { g_qgemv_max_rows = v > 16 ? 16 : (v < 1 ? 1 : v); }

size_t qgemv_workspace_bytes(int64_t, int64_t) { return 0; }  // (the first version quantised the activations into a workspace)

bool qgemv_supported(int wtype, int64_t rows, int64_t K) {
    // whole row segments are fetched with 16-byte loads: every row must start 16-byte aligned (34 * K/32 and 18 * K/32 are multiples of 16 iff
    // K % 256 == 0).  One or two rows: k_qgemv (activation values of a block in registers);  3 .. 16 rows: k_qgemv_rows (rows in LDS as f16,
    // up to 128 KB).  Above that the contraction belongs on the MFMA units.
    if (!(wtype == 8 || wtype == 2) || rows < 1 || rows > g_qgemv_max_rows || K % 256 != 0 || K < 256) return false;
    if (rows <= 2) return K <= 12288;  // 2 rows x 12288 halfs = 48 KB
    const int64_t r = rows <= 4 ? 4 : rows <= 8 ? 8 : 16;
    return r * K * 2 <= 128 * 1024;
}

// x: f32 rows (row stride xs floats, 16-byte aligned), multiplied by pre_scale before the f16 rounding (ggml_ext_linear's scale)
void launch_qgemv(hipStream_t s, float* dst, int64_t ldd, const float* x, int64_t xs, int64_t rows, const void* wraw, int wtype, int64_t K, int64_t M, void*,
                  const Epilogue& ep, float pre_scale, bool pre_silu) {
    const int64_t nblk  = K / 32;
    const size_t wbytes = (size_t)M * (size_t)nblk * (wtype == 8 ? 34 : 18);
    KScope ks_(s, KF_QGEMM, 2.0 * rows * K * M, (double)wbytes + (double)rows * K * 4.0 + (double)rows * M * 4.0);
    QGArgs g;
    g.W         = (const char*)wraw;
    g.row_bytes = nblk * (wtype == 8 ? 34 : 18);
    g.x = x; g.xs = xs;
    g.dst = dst; g.ldd = ldd;
    g.bias = ep.bias; g.residual = ep.residual; g.scale = ep.scale; g.pre_scale = pre_scale;
    g.K = (int)K; g.M = (int)M; g.rows = (int)rows; g.pre_silu = pre_silu ? 1 : 0;
    if (rows > 2) {
        constexpr int CPW2 = 2;
        const unsigned grid2 = (unsigned)((M + 4 * CPW2 - 1) / (4 * CPW2));
        const int r          = rows <= 4 ? 4 : rows <= 8 ? 8 : 16;
        const size_t lds     = (size_t)r * K * 2;
#define QGR_LAUNCH(QT_, R_)                                                                                                         \
    do {                                                                                                                            \
        static bool attr_dev_[64] = {false};                                                                                        \
        int dev_ = 0;                                                                                                               \
        (void)hipGetDevice(&dev_);                                                                                                  \
        if (!attr_dev_[dev_ & 63]) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)k_qgemv_rows<QT_, R_, CPW2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); \
            attr_dev_[dev_ & 63] = true;                                                                                            \
        }                                                                                                                           \
        k_qgemv_rows<QT_, R_, CPW2><<<grid2, 256, lds, s>>>(g);                                                                     \
    } while (0)
        if (wtype == 8) {
            if (r == 4) QGR_LAUNCH(8, 4);
            else if (r == 8) QGR_LAUNCH(8, 8);
            else QGR_LAUNCH(8, 16);
        } else {
            if (r == 4) QGR_LAUNCH(4, 4);
            else if (r == 8) QGR_LAUNCH(4, 8);
            else QGR_LAUNCH(4, 16);
        }
#undef QGR_LAUNCH
        return;
    }
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

// Grouped form: n Linears with raw q8_0 / q4_0 weights that read the SAME one or two activation rows (every Modulation of a FLUX forward reads SiLU(vec),
// flux.hpp:381-428) as ONE weight-streaming launch: dst [rows][Mtot] holds the members' outputs side by side (member m at column members[m].start).
// 57 launches of 16-32 MB each never reach the stream rate (launch ramp and tail per launch: 0.7 TB/s inside the FLUX step); one
// launch over the 1.8 GB does.  `members` is a DEVICE table sorted by start; every member's row count must be a multiple of 4.
void launch_qgemv_group(hipStream_t s, float* dst, int64_t Mtot, const float* x, int64_t xs, int64_t rows, const void* members_dev, int n_members, int wtype, int64_t K,
                        float pre_scale, bool pre_silu) {
    const int64_t nblk  = K / 32;
    const size_t wbytes = (size_t)Mtot * (size_t)nblk * (wtype == 8 ? 34 : 18);
    KScope ks_(s, KF_QGEMM, 2.0 * rows * K * Mtot, (double)wbytes + (double)rows * K * 4.0 + (double)rows * Mtot * 4.0);
    QGArgs g;
    g.members   = (const QGMember*)members_dev;
    g.n_members = n_members;
    g.W         = nullptr;
    g.row_bytes = nblk * (wtype == 8 ? 34 : 18);
    g.x = x; g.xs = xs;
    g.dst = dst; g.ldd = Mtot;
    g.bias = nullptr; g.residual = nullptr; g.scale = 1.f; g.pre_scale = pre_scale;
    g.K = (int)K; g.M = (int)Mtot; g.rows = (int)rows; g.pre_silu = pre_silu ? 1 : 0;
    constexpr int CPW = 4;
    const unsigned grid = (unsigned)((Mtot + 4 * CPW - 1) / (4 * CPW));
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
size_t qgemv_member_bytes() { return sizeof(QGMember); }
void qgemv_fill_member(void* host_entry, const void* W, const float* bias, int start) {
    QGMember m{(const char*)W, bias, start, 0};
    memcpy(host_entry, &m, sizeof(m));
}

// =====================================================================================================
// k_qgemm16 — the same raw-block weight stream on the MATRIX CORES, for Linears with a few hundred activation rows (text-stream Linears of the
// DiTs, the text encoders' 77..512 tokens, small latents): below ~300 rows (2.5 PFLOP/s / 8 TB/s) a GEMM against the f16 weight image is
// bound by the 2 B/weight it streams; the raw blocks are 1.06 B (q8_0) / 0.56 B (q4_0) per weight and no image is ever built or kept.
//
//   W   raw GGUF rows.  Per K segment of 8 blocks (256 weights) a wave fetches the contiguous 272 B / 144 B piece of each of ITS 32 columns with
//       coalesced 16-byte loads (17 / 9 consecutive lanes per column piece) into a per-wave LDS strip; every lane then picks the bytes of its
//       MFMA B fragment out of the strip with aligned dword reads (+ v_alignbit: blocks are 2-byte aligned) and dequantises in registers:
//       byte u -> f16 (1024 + u) by a v_perm with the exponent byte 0x64, minus 1152 (q8_0: u = q ^ 0x80) / 1032 (q4_0 nibbles) = the exact
//       integer, times the block scale d with v_pk_mul_f16 = f16(d * q) round-to-nearest: bit-identical to the value the f16 image holds
//       (wgemm.hip wload), so this kernel and k_gemm16 differ by f32 summation order only.
//   A   the f16 operand image [rows][K] every gemm16 Linear uses (written by the producing kernel), staged per segment into LDS by the whole
//       workgroup (16-byte loads, 512-byte runs), rows padded by 16 B so a ds_read_b128 of 16 consecutive rows covers all banks.
//   k order inside a block follows the quantised bytes, not 0..31: lane group kg (= lane / 32) of v_mfma_f32_32x32x16_f16 step s multiplies
//       q8_0: weights 16 kg + 8 s + i (one contiguous 16-byte run per lane and block);  q4_0: 16 s + 8 kg + i (low nibbles of bytes
//       8 kg .. 8 kg + 7 are elements 8 kg + i, the high nibbles elements 16 + 8 kg + i);  the A fragment is read at the same offsets.
//   Tile: 32 RB rows x 128 columns (4 waves x 32 columns, every wave all rows), next segment's global loads in flight (registers) during the
//   MFMAs of the current one.  Optional split-K over gridDim.z into f32 slabs (combined by k_splitk_reduce) for shapes with few tiles.
// =====================================================================================================
struct QG16Args {
    const char* W;
    int64_t row_bytes;
    const _Float16* A;
    int64_t lda;  // halfs
    float* dst;
    int64_t ldd;
    _Float16* dst16;  // gelu != 0: f16 output rows (operand image of the next Linear), row stride ldd16
    int64_t ldd16;
    const float* bias;
    const float* residual;
    const float* gate;
    int gate_L, gelu;
    float scale;
    int R, M;
    int nseg, nseg_slice;
    int64_t slab;  // > 0: split-K, slice z writes acc * scale to dst + z * slab (bias / residual are added by the reduce pass)
    int a_runL, a_runS;  // > 0: A rows in runs (Epilogue::a_run_L / a_run_S)
};

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// four bytes of u -> two half2 {1024 + byte0, 1024 + byte1}, {1024 + byte2, 1024 + byte3}, minus off, times d
__device__ __forceinline__ void qg_deq4(uint32_t u, half2_t off, half2_t d2, uint32_t& o01, uint32_t& o23) {
    const uint32_t p01 = __builtin_amdgcn_perm(0x64646464u, u, 0x04010400u);
    const uint32_t p23 = __builtin_amdgcn_perm(0x64646464u, u, 0x04030402u);
    const half2_t h01  = (__builtin_bit_cast(half2_t, p01) - off) * d2;
    const half2_t h23  = (__builtin_bit_cast(half2_t, p23) - off) * d2;
    o01                = __builtin_bit_cast(uint32_t, h01);
    o23                = __builtin_bit_cast(uint32_t, h23);
}

// PF = 2 (round 6): TWO segments of global loads in flight (two register sets, loop unrolled by two).  With one set a workgroup's K loop was a chain of
// nseg x (HBM round trip -> LDS -> dequantise -> MFMA) with ~0.3 us of arithmetic per link: FLUX's 256-token text-stream Linears ran at 0.27 TB/s of weight bytes
// (58 us per launch, 76 launches = 4.4 ms of the step)
template <int QT, int RB, int PF = 1>
__global__ __launch_bounds__(256) void k_qgemm16(QG16Args g) {
    constexpr int BLK = QT == 8 ? 34 : 18;
    constexpr int SEG = 8;               // blocks per K segment (256 weights)
    constexpr int CS  = SEG * BLK;       // bytes of one column piece: 272 / 144 (multiples of 16)
    constexpr int NGC = CS / 16;         // 16-byte granules per column piece: 17 / 9
    constexpr int NGW = 32 * NGC;        // granules of a wave's strip: 544 / 288
    constexpr int NLW = (NGW + 63) / 64; // loads per lane: 9 / 5
    constexpr int RT  = RB * 32;
    constexpr int AS  = 256 * 2 + 16;    // bytes between A rows in LDS
    constexpr int NLA = RT * 32 / 256;   // A granules per thread
    constexpr int NB  = QT == 8 ? 16 : 8;  // quant bytes per lane and block
    __shared__ __attribute__((aligned(16))) char As[RT * AS];
    __shared__ __attribute__((aligned(16))) char Wsm[4][32 * CS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, kg = lane >> 5;
    char* Ws             = Wsm[wave];
    const int64_t row0   = (int64_t)blockIdx.x * RT;
    const int colw       = (int)blockIdx.y * 128 + wave * 32;  // first column of this wave
    const int seg0       = (int)blockIdx.z * g.nseg_slice;
    const int seg1       = min(g.nseg, seg0 + g.nseg_slice);

    float16_t acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

    u32x4_t wreg[NLW], areg[NLA];
    u32x4_t wreg2[PF > 1 ? NLW : 1], areg2[PF > 1 ? NLA : 1];  // second register set (PF = 2)
    auto a_phys = [&](int64_t r) -> int64_t {  // rows in runs (token slices of a wider operand image)
        return g.a_runL > 0 ? (int64_t)((uint32_t)r / (uint32_t)g.a_runL) * g.a_runS + (uint32_t)r % (uint32_t)g.a_runL : r;
    };
    auto fetch_into = [&](int seg, u32x4_t* wr, u32x4_t* ar) {
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const int idx = i * 64 + lane;
            const int col = idx / NGC, gr = idx - col * NGC;
            const int cc  = min(colw + col, g.M - 1);
            wr[i]         = (u32x4_t){0, 0, 0, 0};
            if (idx < NGW) wr[i] = *(const u32x4_t*)(g.W + (int64_t)cc * g.row_bytes + (int64_t)seg * CS + gr * 16);
        }
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int idx     = i * 256 + (int)threadIdx.x;
            const int row     = idx >> 5, gr = idx & 31;
            const int64_t grw = row0 + row;
            ar[i]             = (u32x4_t){0, 0, 0, 0};
            if (grw < g.R) ar[i] = *(const u32x4_t*)(g.A + a_phys(grw) * g.lda + (int64_t)seg * 256 + gr * 8);
        }
    };
    auto stash = [&](const u32x4_t* wr, const u32x4_t* ar) {  // registers -> LDS (between two workgroup barriers)
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const int idx = i * 64 + lane;
            if (idx < NGW) *(u32x4_t*)(Ws + idx * 16) = wr[i];  // strip = [column][piece bytes]: idx * 16 = col * CS + gr * 16
        }
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int idx = i * 256 + (int)threadIdx.x;
            *(u32x4_t*)(As + (idx >> 5) * AS + (idx & 31) * 16) = ar[i];
        }
    };
    auto fetch = [&](int seg) {
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const int idx = i * 64 + lane;
            const int col = idx / NGC, gr = idx - col * NGC;
            const int cc  = min(colw + col, g.M - 1);
            wreg[i]       = (u32x4_t){0, 0, 0, 0};
            if (idx < NGW) wreg[i] = *(const u32x4_t*)(g.W + (int64_t)cc * g.row_bytes + (int64_t)seg * CS + gr * 16);
        }
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int idx     = i * 256 + (int)threadIdx.x;
            const int row     = idx >> 5, gr = idx & 31;
            const int64_t grw = row0 + row;
            areg[i]           = (u32x4_t){0, 0, 0, 0};
            if (grw < g.R) areg[i] = *(const u32x4_t*)(g.A + a_phys(grw) * g.lda + (int64_t)seg * 256 + gr * 8);
        }
    };
    auto compute_seg = [&]() {
#pragma unroll
        for (int b = 0; b < SEG; ++b) {
            const char* blk  = Ws + n * CS + BLK * b;
            const _Float16 d = *(const _Float16*)blk;
            const half2_t d2 = {d, d};
            const int ph = (BLK * b + 2) & 2;  // compile-time after unrolling: phase of this lane's quant bytes inside a dword (CS, NB * kg are multiples of 4)
            const char* qp = blk + 2 + NB * kg;
            uint32_t q[NB / 4];
            if (ph == 0) {
#pragma unroll
                for (int j = 0; j < NB / 4; ++j) q[j] = *(const uint32_t*)(qp + 4 * j);
            } else {
                uint32_t raw[NB / 4 + 1];
#pragma unroll
                for (int j = 0; j <= NB / 4; ++j) raw[j] = *(const uint32_t*)(qp - 2 + 4 * j);
#pragma unroll
                for (int j = 0; j < NB / 4; ++j) q[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], 16);
            }
            uint32_t f0[4], f1[4];  // B fragments of MFMA steps 0 and 1 (8 halfs each)
            if constexpr (QT == 8) {
                const half2_t off = {(_Float16)1152.f, (_Float16)1152.f};
                qg_deq4(q[0] ^ 0x80808080u, off, d2, f0[0], f0[1]);
                qg_deq4(q[1] ^ 0x80808080u, off, d2, f0[2], f0[3]);
                qg_deq4(q[2] ^ 0x80808080u, off, d2, f1[0], f1[1]);
                qg_deq4(q[3] ^ 0x80808080u, off, d2, f1[2], f1[3]);
            } else {
                const half2_t off = {(_Float16)1032.f, (_Float16)1032.f};
                qg_deq4(q[0] & 0x0F0F0F0Fu, off, d2, f0[0], f0[1]);
                qg_deq4(q[1] & 0x0F0F0F0Fu, off, d2, f0[2], f0[3]);
                qg_deq4((q[0] >> 4) & 0x0F0F0F0Fu, off, d2, f1[0], f1[1]);
                qg_deq4((q[1] >> 4) & 0x0F0F0F0Fu, off, d2, f1[2], f1[3]);
            }
            const half8_t bf0 = __builtin_bit_cast(half8_t, (u32x4_t){f0[0], f0[1], f0[2], f0[3]}), bf1 = __builtin_bit_cast(half8_t, (u32x4_t){f1[0], f1[1], f1[2], f1[3]});
            const int k0 = 32 * b + (QT == 8 ? 16 * kg : 8 * kg);      // step 0
            const int k1 = 32 * b + (QT == 8 ? 16 * kg + 8 : 16 + 8 * kg);  // step 1
            half8_t af0[RB], af1[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const char* ar = As + (rb * 32 + n) * AS;
                af0[rb]        = *(const half8_t*)(ar + k0 * 2);
                af1[rb]        = *(const half8_t*)(ar + k1 * 2);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0[rb], bf0, acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1[rb], bf1, acc[rb], 0, 0, 0);
        }
    };
    if constexpr (PF > 1) {
        if (seg0 < seg1) fetch_into(seg0, wreg, areg);
        if (seg0 + 1 < seg1) fetch_into(seg0 + 1, wreg2, areg2);
        for (int seg = seg0; seg < seg1; seg += 2) {
            __syncthreads();  // every wave is done reading the previous segment
            stash(wreg, areg);
            __syncthreads();
            if (seg + 2 < seg1) fetch_into(seg + 2, wreg, areg);  // two segments in flight during the MFMAs below
            compute_seg();
            if (seg + 1 >= seg1) break;
            __syncthreads();
            stash(wreg2, areg2);
            __syncthreads();
            if (seg + 3 < seg1) fetch_into(seg + 3, wreg2, areg2);
            compute_seg();
        }
    } else {
        if (seg0 < seg1) fetch(seg0);
        for (int seg = seg0; seg < seg1; ++seg) {
            __syncthreads();  // every wave is done reading the previous segment
            stash(wreg, areg);
            __syncthreads();
            if (seg + 1 < seg1) fetch(seg + 1);  // in flight during the MFMAs below
            compute_seg();
        }
    }
    // ---- store: register r of a 32x32 block holds row (r & 3) + 8 (r >> 2) + 4 kg, lanes run along columns
    const int col = colw + n;
    if (col >= g.M) return;
    const float bias = (g.bias && g.slab == 0) ? g.bias[col] : 0.f;
    float* dst       = g.dst + (g.slab > 0 ? (int64_t)blockIdx.z * g.slab : 0);
    const int mode   = g.slab > 0 ? 0 : g.gelu ? 1 : g.gate ? 2 : g.residual ? 3 : 0;  // workgroup-uniform
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int rbase = (int)row0 + rb * 32 + 4 * kg;
        if ((int)row0 + rb * 32 >= g.R) break;
        if (mode == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < g.R) dst[(int64_t)row * g.ldd + col] = acc[rb][r] * g.scale + bias;
            }
        } else if (mode == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < g.R) g.dst16[(int64_t)row * g.ldd16 + col] = (_Float16)act_apply<UN_GELU>(acc[rb][r] * g.scale + bias);
            }
        } else {
            // gate / residual operands of the 16 registers are fetched together from clamped rows (a predicated load per element made the
            // compiler drain vmcnt each time: 16 dependent round trips per block)
            float rv[16], gv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.R - 1);
                rv[r]         = g.residual[(int64_t)row * g.ldd + col];
                gv[r]         = mode == 2 ? g.gate[(int64_t)((uint32_t)row / (uint32_t)g.gate_L) * g.M + col] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < g.R) dst[(int64_t)row * g.ldd + col] = (acc[rb][r] * g.scale + bias) * gv[r] + rv[r];
            }
        }
    }
}

// =====================================================================================================
// k_wswz_q — raw q8_0 / q4_0 rows -> the f16 MFMA weight image of k_gemm16 ([rows/32][K/16][64 lanes][8 halfs], wgemm.hip), FAST: this is the
// "just-in-time image" of the resident-quantised mode (planner option jit_qimages): HBM keeps only the raw GGUF blocks, a Linear's f16 image
// is rebuilt into a shared scratch right in front of its GEMM and read back out of the 256 MB Infinity Cache, never kept.
// One wave per (32 weight rows, 256 k): the 32 contiguous 272 / 144-byte pieces are fetched with coalesced 16-byte loads into a per-wave LDS strip
// (as k_qgemm16 does), every lane then decodes the blocks of ITS row into the two fragments of each block — same byte -> f16 trick, same
// f16(d * q) values as wload() in wgemm.hip, bit for bit — and the wave writes 1 KiB per fragment.
template <int QT>
__global__ __launch_bounds__(256) void k_wswz_q(half8_t* __restrict__ dst, const char* __restrict__ W, int64_t row_bytes, int64_t R, int64_t kfr) {
    constexpr int BLK = QT == 8 ? 34 : 18;
    constexpr int SEG = 8;
    constexpr int CS  = SEG * BLK;
    constexpr int NGC = CS / 16;
    constexpr int NGW = 32 * NGC;
    constexpr int NLW = (NGW + 63) / 64;
    __shared__ __attribute__((aligned(16))) char Wsm[4][32 * CS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hi = lane >> 5;
    char* Ws         = Wsm[wave];
    const int64_t rb = (int64_t)blockIdx.x * 4 + wave;  // 32-row block of the image
    const int seg    = blockIdx.y;
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int idx = i * 64 + lane;
        if (idx < NGW) {
            const int col    = idx / NGC, gr = idx - col * NGC;
            const int64_t rr = min(rb * 32 + col, R - 1);
            *(u32x4_t*)(Ws + idx * 16) = *(const u32x4_t*)(W + rr * row_bytes + (int64_t)seg * CS + gr * 16);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);  // this wave's strip only: no workgroup barrier needed (each wave reads what it wrote)
    __builtin_amdgcn_wave_barrier();
    const bool real = rb * 32 + n < R;  // rows that pad the image to a multiple of 128 are zero
    half8_t* out    = dst + (rb * kfr + (int64_t)seg * 16) * 64 + lane;
#pragma unroll
    for (int b = 0; b < SEG; ++b) {
        const char* blk  = Ws + n * CS + BLK * b;
        const _Float16 d = real ? *(const _Float16*)blk : (_Float16)0.f;
        const half2_t d2 = {d, d};
        // this lane's quant bytes: q8_0 elements 8 hi .. 8 hi + 7 (fragment 2b) and 16 + 8 hi .. (fragment 2b + 1); q4_0 bytes 8 hi .. 8 hi + 7,
        // whose low nibbles are elements 8 hi + i (fragment 2b) and whose high nibbles are elements 16 + 8 hi + i (fragment 2b + 1)
        uint32_t qa[2], qb[2];
        auto rd8 = [&](const char* qp, uint32_t (&q)[2]) {
            if ((((BLK * b + 2) & 2) == 0)) {  // compile-time after unrolling (CS and the run offsets are multiples of 4)
                q[0] = *(const uint32_t*)qp;
                q[1] = *(const uint32_t*)(qp + 4);
            } else {
                const uint32_t r0 = *(const uint32_t*)(qp - 2), r1 = *(const uint32_t*)(qp + 2), r2 = *(const uint32_t*)(qp + 6);
                q[0] = __builtin_amdgcn_alignbit(r1, r0, 16);
                q[1] = __builtin_amdgcn_alignbit(r2, r1, 16);
            }
        };
        uint32_t f0[4], f1[4];
        if constexpr (QT == 8) {
            rd8(blk + 2 + 8 * hi, qa);
            rd8(blk + 2 + 16 + 8 * hi, qb);
            const half2_t off = {(_Float16)1152.f, (_Float16)1152.f};
            qg_deq4(qa[0] ^ 0x80808080u, off, d2, f0[0], f0[1]);
            qg_deq4(qa[1] ^ 0x80808080u, off, d2, f0[2], f0[3]);
            qg_deq4(qb[0] ^ 0x80808080u, off, d2, f1[0], f1[1]);
            qg_deq4(qb[1] ^ 0x80808080u, off, d2, f1[2], f1[3]);
        } else {
            rd8(blk + 2 + 8 * hi, qa);
            const half2_t off = {(_Float16)1032.f, (_Float16)1032.f};
            qg_deq4(qa[0] & 0x0F0F0F0Fu, off, d2, f0[0], f0[1]);
            qg_deq4(qa[1] & 0x0F0F0F0Fu, off, d2, f0[2], f0[3]);
            qg_deq4((qa[0] >> 4) & 0x0F0F0F0Fu, off, d2, f1[0], f1[1]);
            qg_deq4((qa[1] >> 4) & 0x0F0F0F0Fu, off, d2, f1[2], f1[3]);
        }
        out[(2 * b) * 64]     = __builtin_bit_cast(half8_t, (u32x4_t){f0[0], f0[1], f0[2], f0[3]});
        out[(2 * b + 1) * 64] = __builtin_bit_cast(half8_t, (u32x4_t){f1[0], f1[1], f1[2], f1[3]});
    }
}
bool wswz_q_supported(int wtype, int64_t K) { return (wtype == 8 || wtype == 2) && K % 256 == 0 && K >= 256; }
// dst: image of wswz_bytes(R, K) bytes (rows padded to 128); plain (not GEGLU-paired) row order
void launch_wswz_q(hipStream_t s, void* dst, const void* wraw, int wtype, int64_t K, int64_t R) {
    const int64_t Rp = (R + 127) / 128 * 128, nblk = K / 32;
    KScope ks_(s, KF_PACK_F16, 0.0, (double)R * nblk * (wtype == 8 ? 34 : 18) + (double)Rp * K * 2.0);
    const dim3 grid((unsigned)(Rp / 128), (unsigned)(K / 256));
    if (wtype == 8)
        k_wswz_q<8><<<grid, 256, 0, s>>>((half8_t*)dst, (const char*)wraw, nblk * 34, R, K / 16);
    else
        k_wswz_q<4><<<grid, 256, 0, s>>>((half8_t*)dst, (const char*)wraw, nblk * 18, R, K / 16);
}

// option "qgemm16_max_rows": Linears with 5 .. max_rows activation rows take k_qgemm16 (0 = never).  Default 512 = the text streams of the DiTs and the
// text encoders.  Two measurements decide it: (1) the single-Linear probe with L2-warm weights (profiles/r02s_qgemm_paths_probe.txt) has the f16-image
// GEMM 1.3-2.3x faster kernel-for-kernel (3072 -> 9216 q8_0 at 77 rows 48 vs 21 us, at 256 rows 68 vs 32 us); (2) inside FLUX.1-dev, where every
// layer's weights arrive cold from HBM, the 152 text-stream Linears on raw q4_0 blocks cost 0.3 % of the step (125.1 / 125.0 vs 124.5 / 124.8 ms,
// alternating runs on one box) and take 4.3 GB of f16 images out of HBM (17.2 -> 12.9 GB; profiles/r03g_flux_resident_quantised.txt).  The 4096-token image stream stays on the f16 image: with
// qgemm16_max_rows = 8192 the step takes 167 ms (+30 %) for 26 MB of images instead of 12.9 GB — the resident-quantised mode, selectable.
static int g_qg16_max_rows = 512;
void qgemm16_set_max_rows(int v) { g_qg16_max_rows = v; }
static int g_qg16_pf = 1;  // option "qgemm16_pf": segments of global loads in flight per workgroup.  2 measured SLOWER on the FLUX text stream (few-row family 4.87 -> 6.53 ms per forward, gpurun_out/r08e): the chain is bound by the single wave per SIMD walking LDS read -> dequantise -> MFMA, not by the HBM round trip
void qgemm16_set_pf(int v) { g_qg16_pf = v; }
static int g_qg16_rb = 3;  // option "qgemm16_rb": 32-row blocks per workgroup tile (1 / 2 / 4 forced); 0 = by row count only; 3 (default) = by row count, 64-row tiles for small grids (FLUX text stream: 7.0 -> 6.1 ms per forward, profiles/r05d_family_flux_qgemm16_rb.txt)
void qgemm16_set_rb(int v) { g_qg16_rb = v; }

bool qgemm16_supported(int wtype, int64_t rows, int64_t K, int64_t M) {
    // column pieces are fetched with 16-byte loads: rows of the weight must start 16-byte aligned (K % 256 == 0); 1-2 rows belong to k_qgemv
    return (wtype == 8 || wtype == 2) && rows >= 3 && rows <= g_qg16_max_rows && K % 256 == 0 && K >= 256 && M >= 1 && M < (1ll << 30);  // (the planner asks qgemv_supported first)
}

// split-K slices for a launch (1 = none): plain outputs only (the reduce pass adds bias and residual)
int qgemm16_split_k(int64_t rows, int64_t K, int64_t M) {
    const int64_t rt    = rows <= 32 ? 32 : rows <= 64 ? 64 : 128;
    const int64_t tiles = ((rows + rt - 1) / rt) * ((M + 127) / 128);
    const int64_t nseg  = K / 256;
    int64_t S           = 512 / tiles;  // aim for two workgroups per CU
    if (S > nseg / 2) S = nseg / 2;     // at least two segments per slice
    if (S > 16) S = 16;
    return S < 2 ? 1 : (int)S;
}

void launch_qgemm16(hipStream_t s, float* dst, void* dst16, int64_t ldd16, const void* a16, int64_t lda, int64_t rows, const void* wraw, int wtype, int64_t K,
                    int64_t M, const Epilogue& ep, float* splitk_ws, int splitk_S) {
    const int64_t nblk  = K / 32;
    const size_t wbytes = (size_t)M * (size_t)nblk * (wtype == 8 ? 34 : 18);
    KScope ks_(s, KF_QGEMM, 2.0 * rows * K * M, (double)wbytes + (double)rows * K * 2.0 + (double)rows * M * 4.0);
    QG16Args g{};
    g.W = (const char*)wraw;
    g.row_bytes = nblk * (wtype == 8 ? 34 : 18);
    g.A = (const _Float16*)a16; g.lda = lda;
    g.dst = dst; g.ldd = M; g.dst16 = (_Float16*)dst16; g.ldd16 = ldd16;
    g.bias = ep.bias; g.residual = ep.residual; g.gate = ep.gate; g.gate_L = ep.gate_L > 0 ? ep.gate_L : 1; g.gelu = ep.gelu; g.scale = ep.scale;
    g.R = (int)rows; g.M = (int)M;
    if (ep.a_run_L > 0 && ep.a_run_S != ep.a_run_L) {
        g.a_runL = (int)ep.a_run_L;
        g.a_runS = (int)ep.a_run_S;
    }
    g.nseg = (int)(K / 256);
    const int S = (splitk_ws && splitk_S > 1 && !ep.gate && !ep.gelu && dst) ? splitk_S : 1;
    g.nseg_slice = (g.nseg + S - 1) / S;
    if (S > 1) {
        g.slab = rows * M;
        g.dst  = splitk_ws;
    }
    if ((ep.gelu && !dst16) || (!ep.gelu && !dst) || (ep.gate && !ep.residual)) {
        fprintf(stderr, "ggml-mi355x: invalid k_qgemm16 epilogue request\n");
        abort();
    }
    // 128-row tiles re-use a dequantised B fragment four times, but a launch of few tiles (256 text tokens x 3072 -> 12288: 2 x 96 workgroups of four
    // waves = one wave per SIMD on 3/4 of the CUs) cannot hide its load -> LDS -> dequantise -> MFMA chain: 64-row tiles double the workgroups in flight
    int rb          = rows <= 32 ? 1 : rows <= 64 ? 2 : 4;
    if (g_qg16_rb == 1 || g_qg16_rb == 2 || g_qg16_rb == 4) rb = rows <= 32 ? 1 : (rows <= 64 && g_qg16_rb > 2) ? 2 : g_qg16_rb;
    else if (g_qg16_rb == 3 && rb == 4 && ((rows + 127) / 128) * ((M + 127) / 128) * S < 512) rb = 2;  // 3 = auto: 64-row tiles while the grid stays under two workgroups per CU
    const dim3 grid((unsigned)((rows + rb * 32 - 1) / (rb * 32)), (unsigned)((M + 127) / 128), (unsigned)S);
    // two segments in flight (PF = 2) whenever a workgroup walks at least four segments and the register sets fit (RB <= 2: 64-row tiles; option "qgemm16_pf")
    const bool pf2 = g_qg16_pf >= 2 && rb <= 2 && g.nseg_slice >= 4;
#define QG16_LAUNCH(QT_, RB_)                                         \
    do {                                                              \
        if (pf2 && (RB_) <= 2)                                        \
            k_qgemm16<QT_, ((RB_) <= 2 ? (RB_) : 2), 2><<<grid, 256, 0, s>>>(g); \
        else                                                          \
            k_qgemm16<QT_, RB_, 1><<<grid, 256, 0, s>>>(g);           \
    } while (0)
    if (wtype == 8) {
        if (rb == 1) QG16_LAUNCH(8, 1);
        else if (rb == 2) QG16_LAUNCH(8, 2);
        else QG16_LAUNCH(8, 4);
    } else {
        if (rb == 1) QG16_LAUNCH(4, 1);
        else if (rb == 2) QG16_LAUNCH(4, 2);
        else QG16_LAUNCH(4, 4);
    }
#undef QG16_LAUNCH
    if (S > 1) splitk_reduce_rows(s, dst, splitk_ws, S, rows * M, ep.bias, M, ep.residual);
}


// =====================================================================================================
// k_fgemv — Linear with f16 / f32 weights under a handful of activation rows: the time-embedding MLP and the per-ResBlock embedding projections
// (SiLU(emb) -> Linear 1280 -> C on one row per image, block.hpp:126-160 / unet.hpp time_embed) and the DiTs' vector embedders.  On the MFMA
// GEMM such a Linear is four launches (SiLU, f16 pack, a one-workgroup-per-tile GEMM whose K loop is a pure latency chain, split-K reduce)
// of ~6-20 us each for ~3 MB of weights.  Here: ONE launch; the workgroup stages the (optionally SiLU-activated) rows in LDS once, every wave
// streams its weight rows with coalesced 16-byte loads exactly once and keeps R x CPW f32 accumulators.
// Rounding points = the MFMA path's for f16 weights (activation -> f16, f32 products and sums), so a batch that crosses the row limit agrees up
// to summation order; f32 weights keep f32 x f32 like ggml-cpu's vec_dot_f32 (the MFMA path rounds both to f16).
// =====================================================================================================
struct FGArgs {
    const char* W;  // [M][K] rows of f16 or f32
    const float* x;
    int64_t xs;     // floats between activation rows
    float* dst;
    int64_t ldd;
    const float* bias;
    const float* residual;
    float scale;
    int K, M, rows, pre_silu;
};

// Shape of the code matters as much as the data flow here: the instruction cache starts cold at every launch and this kernel runs its code
// about once, so a version fully unrolled over the rows (staging, FMAs and 32 wave reductions for 16 rows: ~30 KB of straight-line code)
// took 6 us at 2 rows and 23 us at 16 rows whatever K and M were (profiles/r02v_fgemv_probe.txt) — instruction fetch, not work.  Now the rows are
// walked four at a time by ROLLED loops (a ~3 KB body that stays in the cache), the weights of the wave's CPW columns stay in registers.
//   NIT = K steps of 512 whose weight loads are issued together, before the rows are even staged: one memory round trip for K <= 512 NIT.
template <bool W16, int CPW, int NIT>
__global__ __launch_bounds__(256) void k_fgemv(FGArgs g) {
    using XT = typename std::conditional<W16, _Float16, float>::type;
    typedef float f32x8_t __attribute__((ext_vector_type(8)));
    using WV = typename std::conditional<W16, half8_t, f32x8_t>::type;  // 8 weights as loaded
    constexpr int RQ = 4;  // rows per pass of the compute loop
    extern __shared__ __attribute__((aligned(16))) char fg_smem[];  // [rows rounded up to RQ][K] XT
    XT* xl         = (XT*)fg_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = (blockIdx.x * 4 + wave) * CPW;
    const int64_t row_bytes = (int64_t)g.K * (W16 ? 2 : 4);
    WV wv[NIT][CPW];
    auto loadw = [&](int kbase) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = kbase + it * 512 + lane * 8;
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int col = min(col0 + c, g.M - 1);
                // unconditional load from a clamped address (a lane past the end re-reads the last 8 weights and multiplies them by zeros below):
                // a predicated load makes the compiler drain vmcnt after every pair
                const int kk = min(k, g.K - 8);
                wv[it][c]    = *(const WV*)(g.W + (int64_t)col * row_bytes + (size_t)kk * (W16 ? 2 : 4));
            }
        }
    };
    loadw(0);
    // rows -> LDS (SiLU applied, rounded to the operand type): flat float4 chunks, four loads in flight per thread and trip
    {
        const int kq = g.K / 4, rows_p = (g.rows + RQ - 1) / RQ * RQ, total = rows_p * kq;
        const float inv_kq = 1.0f / (float)kq;
        for (int base = threadIdx.x; base < total; base += 4 * 256) {
            float4 a[4];
            int off[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = min(base + j * 256, total - 1);
                int row       = (int)((float)idx * inv_kq);
                row += (row + 1) * kq <= idx;  // float reciprocal: off by at most one
                row -= row * kq > idx;
                const int c4 = idx - row * kq;
                off[j]       = row * g.K + c4 * 4;
                a[j]         = *(const float4*)(g.x + (int64_t)min(row, g.rows - 1) * g.xs + c4 * 4);  // rows of the last partial quad: copies, never stored
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 v = a[j];
                if (g.pre_silu) {
                    v.x = act_apply<UN_SILU>(v.x); v.y = act_apply<UN_SILU>(v.y); v.z = act_apply<UN_SILU>(v.z); v.w = act_apply<UN_SILU>(v.w);
                }
                if (base + j * 256 < total) {
                    XT* d = xl + off[j];
                    d[0] = (XT)v.x; d[1] = (XT)v.y; d[2] = (XT)v.z; d[3] = (XT)v.w;
                }
            }
        }
    }
    __syncthreads();
    if (col0 >= g.M) return;
    for (int t0 = 0; t0 < g.rows; t0 += RQ) {
        float acc[CPW * RQ];  // value index c * RQ + r
#pragma unroll
        for (int i = 0; i < CPW * RQ; ++i) acc[i] = 0.f;
        for (int kbase = 0; kbase < g.K; kbase += NIT * 512) {
            if (kbase > 0) loadw(kbase);  // K > 512 NIT (rare): the weights are re-fetched per row quad from L2
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int k = kbase + it * 512 + lane * 8;
                if (kbase + it * 512 < g.K) {  // wave-uniform; lanes past the end read clamped data and contribute zeros
                    const bool live = k < g.K;
                    const int kx    = live ? k : 0;
#pragma unroll
                    for (int r = 0; r < RQ; ++r) {
                        float xv[8];
                        if (W16) {
                            half8_t h = *(const half8_t*)((const _Float16*)xl + (size_t)(t0 + r) * g.K + kx);
                            if (!live) h = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                            for (int j = 0; j < 8; ++j) xv[j] = (float)h[j];
                        } else {
                            float4 a = *(const float4*)((const float*)xl + (size_t)(t0 + r) * g.K + kx), b = *(const float4*)((const float*)xl + (size_t)(t0 + r) * g.K + kx + 4);
                            if (!live) a = b = make_float4(0.f, 0.f, 0.f, 0.f);
                            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
                        }
#pragma unroll
                        for (int c = 0; c < CPW; ++c)
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[c * RQ + r] = fmaf((float)wv[it][c][j], xv[j], acc[c * RQ + r]);
                    }
                }
            }
        }
        if (g.K > NIT * 512) loadw(0);
        constexpr int NV = CPW * RQ, GRP = 64 / NV;
        const float tot = wave_sum_transpose<NV>(acc, lane);
        const int idx = lane / GRP, c = idx / RQ, t = t0 + idx - c * RQ, col = col0 + c;
        if ((lane & (GRP - 1)) == 0 && t < g.rows && col < g.M) {
            float o = tot * g.scale + (g.bias ? g.bias[col] : 0.f);
            if (g.residual) o += g.residual[(int64_t)t * g.ldd + col];
            g.dst[(int64_t)t * g.ldd + col] = o;
        }
    }
}

static int g_fgemv_max_rows = 16;  // option "fgemv_max_rows" (0 = never)
void fgemv_set_max_rows(int v) { g_fgemv_max_rows = v > 16 ? 16 : v; }

// wtype: 0 = f32, 1 = f16 (ggml type ids)
bool fgemv_supported(int wtype, int64_t rows, int64_t K) {
    if (!(wtype == 0 || wtype == 1) || rows < 1 || rows > g_fgemv_max_rows || K % 8 != 0 || K < 8) return false;
    const int64_t r = (rows + 3) / 4 * 4;
    return r * K * (wtype == 1 ? 2 : 4) <= 96 * 1024;  // the staged rows live in LDS
}

void launch_fgemv(hipStream_t s, float* dst, int64_t ldd, const float* x, int64_t xs, int64_t rows, const void* w, int wtype, int64_t K, int64_t M, const Epilogue& ep,
                  bool pre_silu) {
    const int esz = wtype == 1 ? 2 : 4;
    KScope ks_(s, KF_QGEMM, 2.0 * rows * K * M, (double)M * K * esz + (double)rows * K * 4.0 + (double)rows * M * 4.0);
    FGArgs g;
    g.W = (const char*)w; g.x = x; g.xs = xs; g.dst = dst; g.ldd = ldd;
    g.bias = ep.bias; g.residual = ep.residual; g.scale = ep.scale;
    g.K = (int)K; g.M = (int)M; g.rows = (int)rows; g.pre_silu = pre_silu ? 1 : 0;
    constexpr int CPW = 2, NIT = 3;  // 3 x 512 = the 1280-wide embedding in one round trip
    const unsigned grid = (unsigned)((M + 4 * CPW - 1) / (4 * CPW));
    const size_t lds    = (size_t)((rows + 3) / 4 * 4) * K * esz;
    static bool attr_dev_[64][2] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (wtype == 1) {
        if (!attr_dev_[dev_ & 63][0]) {
            (void)hipFuncSetAttribute((const void*)k_fgemv<true, CPW, NIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_dev_[dev_ & 63][0] = true;
        }
        k_fgemv<true, CPW, NIT><<<grid, 256, lds, s>>>(g);
    } else {
        if (!attr_dev_[dev_ & 63][1]) {
            (void)hipFuncSetAttribute((const void*)k_fgemv<false, CPW, NIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_dev_[dev_ & 63][1] = true;
        }
        k_fgemv<false, CPW, NIT><<<grid, 256, lds, s>>>(g);
    }
}

}  // namespace mi355x
