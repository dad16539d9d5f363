// wgemm.hip — the two dense-contraction workhorses of the denoise step and the VAE decoder on gfx950:
//
//   k_linear_mfma   y[tok][m]  = sum_k x[tok][k] * W[m][k]          (ggml MUL_MAT with a static weight;
//                                                                     ggml_ext_linear, ggml_extend.hpp:1008-1040)
//   k_conv2d_mfma   y[n][oc][oh][ow] = sum_{ic,kh,kw} x[n][ic][..][..] * W[oc][ic][kh][kw]
//                                                                    (implicit GEMM: replaces the reference's
//                                                                     IM2COL(F16) + MUL_MAT + CONT(permute) chain,
//                                                                     ggml_ext_conv_2d, ggml_extend.hpp:1131-1171)
//
// Numerics = the reference's rounding points: activations are rounded to
// f16, multiplied with f16 weights, accumulated in f32 — here by v_mfma_f32_32x32x16_f16.
//
// MI355X mapping
//   * 64-wide waves; one wave owns a (128 x 64) [linear: tokens x m] or (64 x NPB*32) [conv: oc x pos]
//     accumulator block = up to 8 MFMA 32x32 tiles (128 acc VGPRs), 4 waves per workgroup.
//   * WEIGHTS never touch LDS: they are pre-swizzled ONCE (weights buffers are immutable) into MFMA
//     fragment order [row/32][k/16][64 lanes][8 halfs], so a wave's A/B fragment is one fully coalesced
//     1 KiB global_load_dwordx4 that lands directly in the operand VGPRs (L2-resident across workgroups).
//   * ACTIVATIONS are f32 in HBM (the graph's tensor type); a tile is loaded with 128-bit coalesced
//     loads, rounded to f16 in registers and staged in LDS as [row][32 k + 8 pad] halfs: the 80-byte row
//     stride makes every ds_read_b128 fragment read bank-conflict-free (16 distinct rows -> 16 distinct
//     4-bank slots).  For the conv the LDS tile is the input HALO patch (rows x (TW+2) x 32 ic) and is
//     reused by all 9 taps: 72 MFMAs per wave between barriers, no im2col buffer ever exists in HBM.
//   * Epilogues fuse bias, the ResBlock time-embedding add, the residual add and an optional activation,
//     and write 128-byte contiguous segments (lanes run along the contiguous output dim).
#include "device_utils.h"
#include "kernels.h"

namespace mi355x {

enum { WT_F32 = 0, WT_F16 = 1, WT_Q4_0 = 2, WT_Q8_0 = 8, WT_BF16 = 30 };

static inline int64_t rup(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

size_t wswz_bytes(int64_t R, int64_t K) { return (size_t)rup(R, 128) * (size_t)rup(K, 64) * 2; }

__device__ __forceinline__ float wload(const char* row, int type, int64_t k) {
    if (type == WT_F16) return __half2float(((const __half*)row)[k]);
    if (type == WT_F32) return ((const float*)row)[k];
    if (type == WT_BF16) return __uint_as_float((uint32_t)((const uint16_t*)row)[k] << 16);
    if (type == WT_Q8_0) {
        const char* blk = row + (k / 32) * 34;
        return __half2float(*(const __half*)blk) * (float)((const int8_t*)(blk + 2))[k & 31];
    }
    const char* blk = row + (k / 32) * 18;  // Q4_0
    const int e     = (int)(k & 31);
    const uint8_t q = ((const uint8_t*)(blk + 2))[e & 15];
    const int v     = e < 16 ? (q & 0xF) : (q >> 4);
    return __half2float(*(const __half*)blk) * (float)(v - 8);
}

// one thread per (fragment, lane): writes 8 halfs
// geglu_inner > 0: GEGLU pairing (gemm16.hip EPI_GEGLU) — 32-row block rb of the image holds, for column tile t = rb/4 and q = rb%4
// (wave column wc = q/2, block cb = q%2), rows cb*inner + (2t + wc)*32 + r of the source: each wave then owns a value block (cb = 0)
// and the gate block of the same 32 output columns (cb = 1).
__global__ void k_wswz_linear(half8_t* __restrict__ dst, const char* __restrict__ src, int type, int64_t K, int64_t R, int64_t row_bytes, int64_t Kp,
                              int64_t total, int64_t geglu_inner) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane     = (int)(i & 63);
    const int64_t frag = i >> 6;
    const int64_t kb = frag % (Kp / 16), rb = frag / (Kp / 16);
    int64_t row = rb * 32 + (lane & 31);
    if (geglu_inner > 0) {
        const int64_t t = rb >> 2, q = rb & 3;
        row = (q & 1) * geglu_inner + (2 * t + (q >> 1)) * 32 + (lane & 31);
        if ((2 * t + (q >> 1)) * 32 >= geglu_inner) row = R;  // beyond the last pair: zero rows
    }
    const int64_t k0  = kb * 16 + (lane >> 5) * 8;
    half8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float f = 0.f;
        if (row < R && k0 + j < K) f = wload(src + row * row_bytes, type, k0 + j);
        v[j] = (_Float16)f;
    }
    dst[i] = v;
}
void launch_wswz_linear(hipStream_t s, void* dst, const void* src, int src_type, int64_t K, int64_t R, int64_t src_row_bytes, int64_t geglu_inner) {
    const int64_t Kp = rup(K, 64), Rp = rup(R, 128);
    const int64_t total = (Rp / 32) * (Kp / 16) * 64;
    k_wswz_linear<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((half8_t*)dst, (const char*)src, src_type, K, R, src_row_bytes, Kp, total, geglu_inner);
}

// conv weight [KW,KH,IC,OC] f16 -> rows OC, k = tap*ICp + ic
__global__ void k_wswz_conv(half8_t* __restrict__ dst, const __half* __restrict__ src, int KW, int KH, int64_t IC, int64_t OC, int64_t ICp, int64_t Kp,
                            int64_t total, int icb_major) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane     = (int)(i & 63);
    const int64_t frag = i >> 6;
    const int64_t kb = frag % (Kp / 16), rb = frag / (Kp / 16);
    const int64_t oc = rb * 32 + (lane & 31);
    const int64_t k0 = kb * 16 + (lane >> 5) * 8;
    half8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t k = k0 + j;
        int64_t tap, ic;
        if (icb_major) {  // gemm16: k = (icb * taps + tap) * 64 + ic % 64
            const int64_t taps = (int64_t)KH * KW, kb = k / 64;
            tap = kb % taps;
            ic  = (kb / taps) * 64 + k % 64;
        } else {  // first-generation kernels: k = tap * ICp + ic
            tap = k / ICp;
            ic  = k % ICp;
        }
        _Float16 f = (_Float16)0.f;
        if (oc < OC && ic < IC) {
            const __half h = src[(oc * IC + ic) * (KH * KW) + tap];
            f              = *(const _Float16*)&h;
        }
        v[j] = f;
    }
    dst[i] = v;
}
void launch_wswz_conv(hipStream_t s, void* dst, const void* src, int64_t KW, int64_t KH, int64_t IC, int64_t OC, bool icb_major) {
    const int64_t ICp = rup(IC, 64), Kp = ICp * KW * KH, Rp = rup(OC, 128);
    const int64_t total = (Rp / 32) * (Kp / 16) * 64;
    k_wswz_conv<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((half8_t*)dst, (const __half*)src, (int)KW, (int)KH, IC, OC, ICp, Kp, total, icb_major ? 1 : 0);
}

struct EpiDev {
    const float* bias;
    const float* residual;
    const float* chan_add;
    float scale;
    int act;
};

// =====================================================================================================
// Linear
// =====================================================================================================
constexpr int LDS_ROW = 40;  // halfs per staged row: 32 k + 8 pad (80 B)

template <int WT, int WM>
__global__ __launch_bounds__(256) void k_linear_mfma(float* __restrict__ dst, const float* __restrict__ x, const half8_t* __restrict__ wswz, int64_t tokens,
                                                     int64_t K, int64_t Kp, int64_t M, int64_t x_stride, int64_t d_stride, EpiDev ep) {
    constexpr int TR = WT * 128;  // tile rows (tokens)
    constexpr int NL = TR / 32;   // float4 loads per thread per chunk
    __shared__ __attribute__((aligned(16))) _Float16 lds[TR * LDS_ROW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wt = wave % WT, wm = wave / WT;
    const int64_t tok0 = (int64_t)blockIdx.x * TR;
    const int64_t m0   = (int64_t)blockIdx.y * (WM * 64);
    const int64_t kfr  = Kp / 16;  // fragments per weight row-block

    float16_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (float16_t){0};

    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    float4 stage[NL];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int64_t tok = tok0 + r0 + 32 * i;
            const int64_t k   = k0 + c4 * 4;
            float4 v          = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tok < tokens && k < K) v = *(const float4*)(x + tok * x_stride + k);
            stage[i] = v;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            half4_t h;
            h[0] = (_Float16)stage[i].x;
            h[1] = (_Float16)stage[i].y;
            h[2] = (_Float16)stage[i].z;
            h[3] = (_Float16)stage[i].w;
            *(half4_t*)&lds[(r0 + 32 * i) * LDS_ROW + c4 * 4] = h;
        }
    };

    const bool wave_active = (m0 + wm * 64) < ((M + 63) / 64) * 64;  // wave-uniform: rows beyond the padded weight do not exist
    const half8_t* wbase   = wswz + ((m0 + wm * 64) / 32) * kfr * 64 + lane;
    const int arow       = wt * 128 + (lane & 31);
    const int acol       = (lane >> 5) * 8;

    gload(0);
    for (int64_t k0 = 0; k0 < Kp; k0 += 32) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (k0 + 32 < Kp) gload(k0 + 32);
        if (!wave_active) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int64_t kb = k0 / 16 + ks;
            half8_t bw[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bw[mb] = wbase[(mb * kfr + kb) * 64];
            half8_t aa[4];
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) aa[tb] = *(const half8_t*)&lds[(arow + tb * 32) * LDS_ROW + ks * 16 + acol];
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[tb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa[tb], bw[mb], acc[tb][mb], 0, 0, 0);
        }
    }

    // epilogue: D[i=token][j=m]; lane -> m (contiguous), regs -> tokens
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int64_t m = m0 + wm * 64 + mb * 32 + (lane & 31);
        if (m >= M) continue;
        const float bias = ep.bias ? ep.bias[m] : 0.f;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t tok = tok0 + wt * 128 + tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (tok < tokens) {
                    float v = acc[tb][mb][r] * ep.scale + bias;
                    if (ep.residual) v += ep.residual[tok * d_stride + m];
                    if (ep.act >= 0) v = act_dyn(ep.act, v);
                    dst[tok * d_stride + m] = v;
                }
            }
        }
    }
}

void launch_linear_mfma(hipStream_t s, float* dst, const float* x, const void* wswz, int64_t tokens, int64_t K, int64_t M, int64_t x_stride,
                        int64_t d_stride, const Epilogue& e) {
    EpiDev ep{e.bias, e.residual, e.chan_add, e.scale, e.act};
    const int64_t Kp = rup(K, 64);
    // pick the workgroup shape: wide-m tiles for wide outputs (activation tile re-read M/tile_m times),
    // tall-token tiles for narrow outputs.  All variants share the (128 tok x 64 m) wave tile.
    const int64_t mt64 = (M + 63) / 64;
    if (mt64 % 4 == 0 && tokens >= 128) {
        dim3 grid((unsigned)((tokens + 127) / 128), (unsigned)(mt64 / 4));
        k_linear_mfma<1, 4><<<grid, 256, 0, s>>>(dst, x, (const half8_t*)wswz, tokens, K, Kp, M, x_stride, d_stride, ep);
    } else if (mt64 % 2 == 0 && tokens >= 256) {
        dim3 grid((unsigned)((tokens + 255) / 256), (unsigned)(mt64 / 2));
        k_linear_mfma<2, 2><<<grid, 256, 0, s>>>(dst, x, (const half8_t*)wswz, tokens, K, Kp, M, x_stride, d_stride, ep);
    } else if (tokens > 256) {
        dim3 grid((unsigned)((tokens + 511) / 512), (unsigned)mt64);
        k_linear_mfma<4, 1><<<grid, 256, 0, s>>>(dst, x, (const half8_t*)wswz, tokens, K, Kp, M, x_stride, d_stride, ep);
    } else {
        // few tokens (time-embedding MLPs, cross-attention K/V on 77 tokens): one 128-token tile, 4 m-blocks per WG
        dim3 grid((unsigned)((tokens + 127) / 128), (unsigned)((mt64 + 3) / 4));
        k_linear_mfma<1, 4><<<grid, 256, 0, s>>>(dst, x, (const half8_t*)wswz, tokens, K, Kp, M, x_stride, d_stride, ep);
    }
}

// =====================================================================================================
// Implicit-GEMM conv2d (3x3 pad 1 stride 1|2, 1x1), NCHW f32 in / out
// =====================================================================================================
struct ConvArgs {
    int W, H, IC, N, OC, OW, OH;   // W,H: dims of x as stored; with UPS the conv sees (2W x 2H)
    int ICp;                       // IC padded to 32
    int TW, TH, lgTW;              // output tile (TW pow2)
    int PW, PH;                    // LDS halo patch dims
    int tiles_x, tiles_y;
    int pad;
    int64_t kfr;                   // weight fragments per oc row-block = Kp/16
    unsigned magic_np, magic_pw;   // floor(2^32/d)+1: exact n/d by __umulhi for the small ranges used here
};

template <int NPB, int KS, int S, bool UPS>
__global__ __launch_bounds__(256) void k_conv2d_mfma(float* __restrict__ dst, const float* __restrict__ x, const half8_t* __restrict__ wswz, ConvArgs g, EpiDev ep) {
    extern __shared__ __attribute__((aligned(16))) _Float16 patch[];  // [PH*PW][LDS_ROW]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    int tile       = blockIdx.x;
    const int txi  = tile % g.tiles_x;
    tile /= g.tiles_x;
    const int tyi = tile % g.tiles_y;
    const int n   = tile / g.tiles_y;
    const int tx0 = txi * g.TW, ty0 = tyi * g.TH;
    const int oc0 = blockIdx.y * 64;
    const int iw0 = tx0 * S - g.pad, ih0 = ty0 * S - g.pad;
    const int CW = UPS ? g.W * 2 : g.W, CH = UPS ? g.H * 2 : g.H;  // conv-visible input dims

    float16_t acc[2][NPB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NPB; ++b) acc[a][b] = (float16_t){0};

    // per-lane LDS base (in rows of LDS_ROW) of each of this wave's position blocks, tap (0,0)
    int pbase[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const int j  = (wave * NPB + pb) * 32 + (lane & 31);
        const int ty = j >> g.lgTW, tx = j & (g.TW - 1);
        pbase[pb]    = (ty * S) * g.PW + tx * S;
    }
    const int kcol        = (lane >> 5) * 8;
    const half8_t* wbase  = wswz + (int64_t)(oc0 / 32) * g.kfr * 64 + lane;
    const int npatch      = g.PH * g.PW;
    const int64_t plane   = (int64_t)g.W * g.H;
    const float* xn       = x + (int64_t)n * g.IC * plane;

    for (int c0 = 0; c0 < g.ICp; c0 += 32) {
        __syncthreads();
        // ---- stage the halo patch of 32 input channels: f32 -> f16, [pos][ic]
        for (int e = threadIdx.x; e < npatch * 8; e += 256) {
            const int q  = (int)__umulhi((unsigned)e, g.magic_np);  // ic quad
            const int pp = e - q * npatch;
            const int py = (int)__umulhi((unsigned)pp, g.magic_pw), px = pp - py * g.PW;
            const int ih = ih0 + py, iw = iw0 + px;
            half4_t h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (ih >= 0 && ih < CH && iw >= 0 && iw < CW) {
                const int sy = UPS ? (ih >> 1) : ih, sx = UPS ? (iw >> 1) : iw;
                const float* p = xn + (int64_t)(c0 + q * 4) * plane + (int64_t)sy * g.W + sx;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + q * 4 + j < g.IC) h[j] = (_Float16)p[j * plane];
            }
            *(half4_t*)&patch[pp * LDS_ROW + q * 4] = h;
        }
        __syncthreads();
        // ---- KS*KS taps x 2 k-steps of 16 ic
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int tap    = kh * KS + kw;
                const int tapoff = kh * g.PW + kw;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int64_t kb = ((int64_t)tap * g.ICp + c0) / 16 + ks;
                    half8_t aw[2];
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob) aw[ob] = wbase[(ob * g.kfr + kb) * 64];
#pragma unroll
                    for (int pb = 0; pb < NPB; ++pb) {
                        const half8_t bx = *(const half8_t*)&patch[(pbase[pb] + tapoff) * LDS_ROW + ks * 16 + kcol];
#pragma unroll
                        for (int ob = 0; ob < 2; ++ob) acc[ob][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[ob], bx, acc[ob][pb], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: D[i=oc][j=pos]; lanes run along ow (contiguous)
    const int64_t oplane = (int64_t)g.OW * g.OH;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const int j  = (wave * NPB + pb) * 32 + (lane & 31);
        const int oh = ty0 + (j >> g.lgTW), ow = tx0 + (j & (g.TW - 1));
        if (oh >= g.OH || ow >= g.OW) continue;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = oc0 + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (oc < g.OC) {
                    const int64_t o = ((int64_t)n * g.OC + oc) * oplane + (int64_t)oh * g.OW + ow;
                    float v         = acc[ob][pb][r] * ep.scale;
                    if (ep.bias) v += ep.bias[oc];
                    if (ep.chan_add) v += ep.chan_add[(int64_t)n * g.OC + oc];
                    if (ep.residual) v += ep.residual[o];
                    if (ep.act >= 0) v = act_dyn(ep.act, v);
                    dst[o] = v;
                }
            }
        }
    }
}

template <int NPB, int KS, int S, bool UPS>
static void conv_launch(hipStream_t s, float* dst, const float* x, const void* wswz, ConvArgs g, const EpiDev& ep) {
    const size_t lds = (size_t)g.PH * g.PW * LDS_ROW * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_conv2d_mfma<NPB, KS, S, UPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid((unsigned)(g.tiles_x * g.tiles_y * g.N), (unsigned)((g.OC + 63) / 64));
    k_conv2d_mfma<NPB, KS, S, UPS><<<grid, 256, lds, s>>>(dst, x, (const half8_t*)wswz, g, ep);
}

void launch_conv2d_mfma(hipStream_t s, float* dst, const float* x, const void* wswz, int64_t W, int64_t H, int64_t IC, int64_t N, int64_t OC,
                        int ksize, int stride, int pad, bool upscale2x, const Epilogue& e) {
    EpiDev ep{e.bias, e.residual, e.chan_add, e.scale, e.act};
    ConvArgs g;
    g.W  = (int)W;
    g.H  = (int)H;
    g.IC = (int)IC;
    g.N  = (int)N;
    g.OC = (int)OC;
    const int CW = upscale2x ? (int)W * 2 : (int)W, CH = upscale2x ? (int)H * 2 : (int)H;
    g.OW  = (CW + 2 * pad - ksize) / stride + 1;
    g.OH  = (CH + 2 * pad - ksize) / stride + 1;
    g.ICp = (int)rup(IC, 64);
    g.pad = pad;
    g.kfr = (int64_t)g.ICp * ksize * ksize / 16;

    // tile selection: positions per workgroup in {128, 256, 512}; bigger tiles amortise the weight
    // fragment stream, smaller ones keep >= ~2 workgroups per CU in flight and fit the LDS patch.
    const int64_t total_pos = (int64_t)g.OW * g.OH * N;
    const int64_t oc_tiles  = (OC + 63) / 64;
    auto patch_bytes = [&](int tile_pos, int& TW, int& TH) {
        // TW: power of two <= tile_pos minimising padded positions, ties -> wider
        int best          = 8;
        int64_t best_cost = INT64_MAX;
        for (int tw = 8; tw <= tile_pos; tw *= 2) {
            const int th       = tile_pos / tw;
            const int64_t cost = (int64_t)rup(g.OW, tw) * rup(g.OH, th);
            if (cost <= best_cost) {
                best_cost = cost;
                best      = tw;
            }
        }
        TW = best;
        TH = tile_pos / best;
        const int PW = (TW - 1) * stride + ksize, PH = (TH - 1) * stride + ksize;
        return (size_t)PW * PH * LDS_ROW * 2;
    };
    int TW = 8, TH = 16, npb = -1;
    for (int c : {4, 2, 1}) {
        const int tile_pos = 128 * c;
        const size_t pb    = patch_bytes(tile_pos, TW, TH);
        const int64_t wgs  = ((total_pos + tile_pos - 1) / tile_pos) * oc_tiles;
        if (pb <= 64 * 1024 && wgs >= 512) {
            npb = c;
            break;
        }
    }
    if (npb < 0) {
        for (int c : {1, 2, 4}) {
            if (patch_bytes(128 * c, TW, TH) <= 96 * 1024) {
                npb = c;
                break;
            }
        }
    }
    if (npb < 0) npb = 1;
    (void)patch_bytes(128 * npb, TW, TH);
    g.TW   = TW;
    g.TH   = TH;
    g.lgTW = 31 - __builtin_clz(TW);
    g.PW   = (TW - 1) * stride + ksize;
    g.PH   = (TH - 1) * stride + ksize;
    g.tiles_x = (g.OW + TW - 1) / TW;
    g.tiles_y = (g.OH + TH - 1) / TH;
    g.magic_np = (unsigned)((1ull << 32) / (unsigned)(g.PH * g.PW) + 1);
    g.magic_pw = (unsigned)((1ull << 32) / (unsigned)g.PW + 1);

#define CONV_CASE(NPB_, KS_, S_, UPS_)                                                          \
    if (npb == NPB_ && ksize == KS_ && stride == S_ && upscale2x == UPS_) {                     \
        conv_launch<NPB_, KS_, S_, UPS_>(s, dst, x, wswz, g, ep);                               \
        return;                                                                                 \
    }
    CONV_CASE(4, 3, 1, false) CONV_CASE(2, 3, 1, false) CONV_CASE(1, 3, 1, false)
    CONV_CASE(4, 3, 1, true) CONV_CASE(2, 3, 1, true) CONV_CASE(1, 3, 1, true)
    CONV_CASE(4, 3, 2, false) CONV_CASE(2, 3, 2, false) CONV_CASE(1, 3, 2, false)
    CONV_CASE(4, 1, 1, false) CONV_CASE(2, 1, 1, false) CONV_CASE(1, 1, 1, false)
#undef CONV_CASE
}

}  // namespace mi355x
