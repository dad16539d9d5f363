// qgemm.hip — Linear layers whose weights are q8_0 / q4_0 GGUF blocks and whose activation has only a few rows (adaLN / modulation
// vectors of the DiTs: 1 row per image).  Such a contraction is a pure WEIGHT STREAM: the
// kernel reads the RAW quantised blocks from HBM exactly once (34 B / 18 B per 32 weights, coalesced 16-byte loads of whole row
// segments), dequantises in registers and never builds the f16 weight image the MFMA GEMMs use (that image is 3.5x the bytes of q4_0).
//
// Arithmetic = ggml-cpu's for a quantised src0 (upstream vec_dot_q8_0_q8_0 / vec_dot_q4_0_q8_0): the activation
// row is quantised to q8_0 blocks first (d = amax / 127 stored as f16, q = round(x / d)), every weight block contributes
// d_w * d_x * sum_i(q_w[i] * q_x[i]) with the integer sum on v_dot4_i32_i8; q4_0 weights are (nibble - 8): the "- 8" is applied as
// - 8 * d_w * d_x * sum_i q_x[i], precomputed per activation block.
//
// Mapping: a wave owns CPW consecutive weight rows (= output features) and walks K in segments of 64 blocks — lane l of the wave owns
// block l of the segment for every row, so the activation blocks of the segment sit in registers (R rows x 10 registers) and are reused
// by all CPW weight rows.  A weight row segment (64 x 34 B = 2176 B, or 64 x 18 B = 1152 B) is fetched with 16-byte loads of the whole
// contiguous range into a per-wave LDS strip and re-read block-wise with aligned 4-byte LDS loads + v_alignbit (the blocks are only
// 2-byte aligned).  Cross-lane sums at the end, bias / residual in the store.
#include "device_utils.h"
#include "kernels.h"
#include "ktime.h"

namespace mi355x {

// ---- activation rows -> private q8 blocks: q [rows][K] int8, d [rows][K/32] f32 (the f16-rounded scale), s8 [rows][K/32] = 8 * d * sum(q)
__global__ void k_quant_q8_rows(int8_t* __restrict__ q, float* __restrict__ d, float* __restrict__ s8, const float* __restrict__ x, int64_t xs, int rows, int K,
                                float pre_scale) {
    const int nblk = K / 32;
    const int i    = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nblk) return;
    const int r = i / nblk, b = i - r * nblk;
    const float4* xp = (const float4*)(x + (int64_t)r * xs + b * 32);
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 t = xp[j];
        v[4 * j] = t.x * pre_scale; v[4 * j + 1] = t.y * pre_scale; v[4 * j + 2] = t.z * pre_scale; v[4 * j + 3] = t.w * pre_scale;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[4 * j]), fabsf(v[4 * j + 1])), fmaxf(fabsf(v[4 * j + 2]), fabsf(v[4 * j + 3]))));
    }
    const float dd = amax / 127.f;
    const float id = dd != 0.f ? 1.f / dd : 0.f;
    int sum = 0;
    uint32_t packed[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qi = (int)roundf(v[4 * j + e] * id);
            sum += qi;
            w |= ((uint32_t)(qi & 0xFF)) << (8 * e);
        }
        packed[j] = w;
    }
    uint4* qp = (uint4*)(q + (int64_t)r * K + b * 32);
    qp[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    qp[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
    const float df = (float)(_Float16)dd;  // block_q8_0.d is an f16
    d[i]  = df;
    s8[i] = 8.f * df * (float)sum;
}

struct QGArgs {
    const char* W;       // raw quantised rows
    int64_t row_bytes;
    const int8_t* xq;    // [R][K]
    const float* xd;     // [R][nblk]
    const float* xs8;    // [R][nblk]
    float* dst;
    int64_t ldd;
    const float* bias;
    const float* residual;  // same layout as dst
    float scale;
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
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* my       = strip[wave];
    const int nblk = g.K / 32;
    const int nseg = (nblk + 63) / 64;
    const int col0 = (blockIdx.x * 4 + wave) * CPW;
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
        // activation blocks of this segment: registers, shared by all CPW weight rows
        uint32_t xq[R][8];
        float xd[R], xs8[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int tt = t < g.rows ? t : 0;
            if (have) {
                const uint4* p = (const uint4*)(g.xq + (int64_t)tt * g.K + (int64_t)blk * 32);
                const uint4 a = p[0], b = p[1];
                xq[t][0] = a.x; xq[t][1] = a.y; xq[t][2] = a.z; xq[t][3] = a.w;
                xq[t][4] = b.x; xq[t][5] = b.y; xq[t][6] = b.z; xq[t][7] = b.w;
                xd[t]  = t < g.rows ? g.xd[(int64_t)tt * nblk + blk] : 0.f;
                xs8[t] = t < g.rows ? g.xs8[(int64_t)tt * nblk + blk] : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xq[t][j] = 0;
                xd[t] = xs8[t] = 0.f;
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
                    int isum = 0;
                    if (QT == 8) {
#pragma unroll
                        for (int m = 0; m < 8; ++m) isum = __builtin_amdgcn_sdot4((int)qd[m], (int)xq[t][m], isum, false);
                        acc[c][t] += dw * xd[t] * (float)isum;
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            isum = __builtin_amdgcn_sdot4((int)(qd[m] & 0x0F0F0F0Fu), (int)xq[t][m], isum, false);
                            isum = __builtin_amdgcn_sdot4((int)((qd[m] >> 4) & 0x0F0F0F0Fu), (int)xq[t][m + 4], isum, false);
                        }
                        acc[c][t] += dw * (xd[t] * (float)isum - xs8[t]);
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

size_t qgemv_workspace_bytes(int64_t rows, int64_t K) { return (size_t)rows * K + 2 * (size_t)rows * (K / 32) * sizeof(float) + 64; }

bool qgemv_supported(int wtype, int64_t rows, int64_t K) {
    // whole row segments are fetched with 16-byte loads: every row must start 16-byte aligned (34 * K/32 and 18 * K/32 are multiples of 16 iff K % 256 == 0)
    return (wtype == 8 || wtype == 2) && rows >= 1 && rows <= 4 && K % 256 == 0 && K >= 256;
}

// x: f32 rows (row stride xs floats), pre-multiplied by pre_scale before quantisation (ggml_ext_linear's scale); ws: qgemv_workspace_bytes(rows, K)
void launch_qgemv(hipStream_t s, float* dst, int64_t ldd, const float* x, int64_t xs, int64_t rows, const void* wraw, int wtype, int64_t K, int64_t M, void* ws,
                  const Epilogue& ep, float pre_scale) {
    const int64_t nblk = K / 32;
    int8_t* xq = (int8_t*)ws;
    float* xd  = (float*)((char*)ws + (((size_t)rows * K + 15) & ~(size_t)15));
    float* xs8 = xd + rows * nblk;
    const size_t wbytes = (size_t)M * (size_t)nblk * (wtype == 8 ? 34 : 18);
    KScope ks_(s, KF_QGEMM, 2.0 * rows * K * M, (double)wbytes + (double)rows * K * 4.0 + (double)rows * M * 4.0);
    k_quant_q8_rows<<<(unsigned)((rows * nblk + 127) / 128), 128, 0, s>>>(xq, xd, xs8, x, xs, (int)rows, (int)K, pre_scale);
    QGArgs g;
    g.W         = (const char*)wraw;
    g.row_bytes = nblk * (wtype == 8 ? 34 : 18);
    g.xq = xq; g.xd = xd; g.xs8 = xs8;
    g.dst = dst; g.ldd = ldd;
    g.bias = ep.bias; g.residual = ep.residual; g.scale = ep.scale;
    g.K = (int)K; g.M = (int)M; g.rows = (int)rows;
    constexpr int CPW = 4;
    const unsigned grid = (unsigned)((M + 4 * CPW - 1) / (4 * CPW));
#define QG_LAUNCH(QT_, R_) k_qgemv<QT_, R_, CPW><<<grid, 256, 0, s>>>(g)
    if (wtype == 8) {
        if (rows == 1) QG_LAUNCH(8, 1);
        else if (rows == 2) QG_LAUNCH(8, 2);
        else QG_LAUNCH(8, 4);
    } else {
        if (rows == 1) QG_LAUNCH(4, 1);
        else if (rows == 2) QG_LAUNCH(4, 2);
        else QG_LAUNCH(4, 4);
    }
#undef QG_LAUNCH
}

}  // namespace mi355x
