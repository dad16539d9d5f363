// g16_common.h — argument block and epilogues shared by the gemm16 kernels (gemm16.hip) and the LDS-window conv kernel (conv3w.hip)
#pragma once
#include "device_utils.h"
#include "kernels.h"

namespace mi355x {

struct G16Epi {  // dst = acc * scale + bias (+ residual); DiT variants: (acc*scale + bias) * gate[image][col] + residual, and gelu -> f16
    const float* bias;
    const float* residual;
    float scale;
    const float* gate = nullptr;  // [images][C] per-image per-column gate (adaLN gate_msa / gate_mlp, mmdit.hpp:540-551); needs residual
    int gate_L        = 0;        // rows per image (>= 32)
    int gelu          = 0;        // f16-only output: tanh-GELU before rounding (Mlp fc1 -> fc2, block.hpp:249-258)
    const float* chan_add = nullptr;  // conv only: [N][OC] value added per (image, channel) — the ResBlock's time-embedding ADD (block.hpp:150-160)
    int chan_ld           = 0;        // floats between images of chan_add (0 = C)
};

struct G16Args {
    const _Float16* A;
    int64_t lda;       // halfs (rows mode)
    const half8_t* W;
    int64_t kfr;       // fragments per 32-col block = Kp/16
    float* dst;
    _Float16* dst16;   // optional f16 row-major copy of the output (rows mode), row stride ldd16 halfs
    int64_t ldd, ldd16;
    int geglu_inner;       // > 0: GEGLU epilogue (weights in the paired order of k_wswz_linear), f16 output [rows][geglu_inner]
    int geglu16;           // with geglu_inner > 0: the weight image is in the 16-column interleave (k_wswz_linear, geglu_inner < 0) -> epi_geglu16 (any tile geometry)
    int hm_d, hm_H, hm_L;  // head-major store (rows mode): element (row = n*L + l, col = h*d + dd) -> ((n*H + h)*L + l)*d + dd
    int64_t R, C;
    int64_t row_base;  // rows mode: this launch's tiles start at row row_base (a multiple of 256) of the R rows — the second launch of a row-split Linear (g16_launch: tail rows on smaller tiles)
    int worder;        // > 0: WEIGHT-MAJOR workgroup order, value = row tiles of the launch (g16_wg_order).  Launches whose weight bytes dominate their activation
                       // bytes (the 8x8 / 16x16 UNet levels: 29.5 .. 59 MB of weights against 2.6 .. 10.5 MB of NHWC image, K cut into slices) keep every
                       // (column tile, K slice) weight chunk on ONE XCD — its row tiles run back to back there and re-read the chunk from that XCD's L2 — instead
                       // of the default order (consecutive ids of one XCD share the A rows), under which every XCD fetched every weight chunk:
                       // 297 MB fetched per launch for 29.5 MB of weights (profiles/r06c_pmc_conv256_by_shape.txt)
    int a_runL, a_runS;  // > 0 (rows mode): A rows come in runs — row r is image row (r / a_runL) * a_runS + r % a_runL (Epilogue::a_run_L / a_run_S)
    int qt;            // 8 / 4: W points at RAW GGUF q8_0 / q4_0 rows (qrow_bytes apart), dequantised inside the main loop (k_gemm16<..., QT>: pipelined 256 x 256 tile only); 0: f16 weight image
    int64_t qrow_bytes;
    int wblk_lim;      // > 0: 32-column blocks the weight image holds (columns padded to 128): tiles wider than the padding (256-column tiles on M % 256 == 128) fetch block wblk_lim - 1 instead of reading past the image
    int nt;            // K tiles (BK each)
    int ncol_tiles;
    int split_k;       // > 1: blockIdx.y = K slice; slice s accumulates K tiles [s*nt_slice, ...) into dst + s*slab (raw partial sums)
    int nt_slice;
    int64_t slab;      // elements between the partial-sum slabs
    // in-kernel reduction (sk_cnt != nullptr): every K slice dumps its accumulators to sk_slab[(tile * split_k + slice)] (BM x BN floats, register
    // order), takes a ticket on sk_cnt[tile], and the LAST arriver sums all slices in slice order and runs the regular epilogue
    float* sk_slab;
    int* sk_cnt;
    // stream-K (k_gemm16<..., SK = true>): sk_grid workgroups share sk_tiles * nt (tile, K-tile) units; slab slots 2w / 2w + 1 of workgroup w in sk_slab, one counter per tile in sk_cnt
    int sk_grid, sk_tiles, sk_dp;  // sk_dp: tiles [0, sk_dp) stay whole (sk_dp / sk_grid per workgroup), tiles [sk_dp, sk_tiles) are cut over K (0: every tile may be cut)
    // column tiles with col0 >= split_col (> 0) store gelu(acc * scale + bias) as f16 rows into split_dst16 (pointer pre-offset so that the GLOBAL column indexes it) instead of the f32 output
    int split_col;
    _Float16* split_dst16;
    int64_t split_ldd16;
    // sibling Linears sharing the A operand in ONE launch (rows mode): column tile t belongs to weight t / ncol_tiles; each weight has its own image,
    // destination(s) and bias, everything else (shape, head-major parameters, scale) is common.  multi <= 1: off.
    int multi;
    const half8_t* Wm[16];
    float* dstm[16];
    _Float16* dst16m[16];
    const float* biasm[16];
    // conv gather
    int H, Wd, ICp, OH, OW, S, pad, UPS, KS, icb_per_tap, tap_major;
    int64_t OHOW;
    const _Float16* zero;
    G16Epi ep;
#ifdef MI355X_EXPERIMENTS
    int abl;  // option "gemm16_abl" (wrong-result timing ablations): 5 = GEGLU epilogue without the GELU arithmetic, 6 = epilogue without its stores, 7 = no main loop
#endif
};

// host side (gemm16.hip): row tiles for G16Args::worder when a conv launch of gx x ny workgroups should run weight-major, else 0 (option "conv_wmajor")
int gemm16_worder_rows(const G16Args& g, unsigned gx, unsigned ny);
void gemm16_set_conv_wmajor(int v);
// workgroup -> (tile id, K slice).  Default: XCD-aware tile order over blockIdx.x (consecutive ids on one XCD share the A row tile), slice = blockIdx.y.
// Weight-major (g.worder = row tiles; needs gridDim.x % 8 == 0 and (column tiles x slices) % 8 == 0, checked by the launcher): workgroup L = y * gridDim.x + x
// runs on XCD L % 8; XCD k takes the (column tile, slice) pairs p = k, k + 8, ... and walks the row tiles of one pair back to back.
__device__ __forceinline__ void g16_wg_order(const G16Args& g, int ncol_all, int& bid, int& kslice) {
    if (g.worder > 0) {
        const int L  = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        const int k  = L & 7, j = L >> 3;
        const int pl = j / g.worder, r = j - pl * g.worder;
        const int p  = pl * 8 + k;
        kslice       = p / ncol_all;
        bid          = r * ncol_all + (p - kslice * ncol_all);
        return;
    }
    const int nb = gridDim.x;
    bid          = blockIdx.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    kslice = blockIdx.y;
}

#define GLDS16(gptr, ldsptr) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)


// =====================================================================================================
// epilogue.  The accumulators of one 32x32 block live in 16 registers per lane.  The stores used to be one 64-way unrolled loop
// with the activation switch and every optional pointer tested per element: ~16k instructions per kernel of which a wave
// executed a sparse ~5 %, every step an instruction-cache miss (+50 us per workgroup, profiles/r01b).  Now the launch-/workgroup-
// uniform case analysis happens ONCE and selects a compact straight-line variant: per element one v_fma and one global_store
// with an SGPR base (uniform 64-bit tile pointer + register row) and a single 32-bit per-lane byte offset.
// =====================================================================================================
template <typename T>
__device__ __forceinline__ void st_u(T* ubase, uint32_t lane_bytes, T v) { *(T*)((char*)ubase + lane_bytes) = v; }
template <typename T>
__device__ __forceinline__ T ld_u(const T* ubase, uint32_t lane_bytes) { return *(const T*)((const char*)ubase + lane_bytes); }

enum { EPI_F32 = 0, EPI_F32_RES = 1, EPI_F16 = 2, EPI_HM_F32 = 3, EPI_HM_F16 = 4, EPI_GENERIC = 5, EPI_F16_GELU = 6, EPI_F32_GATE = 7, EPI_F16_RES = 8 };

// FF1 + GEGLU: the weight image is laid out (k_wswz_linear, geglu_inner > 0) so that inside every 128-column tile the 32-column blocks
// are [value w0 | gate w0 | value w1 | gate w1]: a wave's even column block holds 32 value columns, the next odd one the matching gates
template <int RB, int CB>
__device__ __forceinline__ void epi_geglu(const float16_t (&acc)[RB][CB], const G16Args& g, int64_t row0, int col0, int wr, int wc, int lane) {
    static_assert(CB % 2 == 0, "GEGLU pairing needs an even number of column blocks per wave");
    const int hi = lane >> 5, lc = lane & 31;
#pragma unroll
    for (int p = 0; p < CB / 2; ++p) {
        const int gb  = col0 / 32 + wc * CB + 2 * p;  // global 32-column block of the value half
        const int oc0 = ((gb >> 2) * 2 + ((gb & 3) >> 1)) * 32;  // first output column of this pair
        if (oc0 >= g.geglu_inner) continue;
        const float bx = g.ep.bias ? g.ep.bias[oc0 + lc] : 0.f, bg = g.ep.bias ? g.ep.bias[g.geglu_inner + oc0 + lc] : 0.f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int64_t base_row = row0 + wr * (RB * 32) + rb * 32;
            if (base_row >= g.R) continue;
            const int nvl     = (int)(g.R - base_row < 32 ? g.R - base_row : 32) - 4 * hi;
            _Float16* ub      = g.dst16 + base_row * g.ldd16 + oc0;
            const uint32_t lb = (uint32_t)(lc + 4 * hi * (int)g.ldd16) * 2u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (ro >= nvl) continue;
                const float xv = acc[rb][2 * p][r] * g.ep.scale + bx, gv = acc[rb][2 * p + 1][r] * g.ep.scale + bg;
#ifdef MI355X_EXPERIMENTS
                if (g.abl == 5) {
                    st_u(ub + (int64_t)ro * g.ldd16, lb, (_Float16)(xv * gv));
                    continue;
                }
                if (g.abl == 6) {
                    const float y = xv * act_apply<UN_GELU>(gv);
                    if (y == 123456.789f) st_u(ub + (int64_t)ro * g.ldd16, lb, (_Float16)y);
                    continue;
                }
#endif
                st_u(ub + (int64_t)ro * g.ldd16, lb, (_Float16)(xv * act_apply<UN_GELU>(gv)));
            }
        }
    }
}

// FF1 + GEGLU for ANY tile geometry (an odd number of column blocks per wave: the 256 x 160 and 256 x 320 tiles): the weight image interleaves value and
// gate at 16 columns — columns 0..15 of every 32-column block are the values of outputs 16 b .. 16 b + 15, columns 16..31 their gates — so value and gate
// of one output sit in lanes l and l ^ 16 of the same accumulator register.  ONE v_permlane16_swap per register pair (j, j + 8) brings them together:
// swap(vdst = acc[j + 8], src = acc[j]) exchanges vdst's lanes 16..31 with src's lanes 0..15 (and 48..63 with 32..47): afterwards vdst = {values of
// register j + 8 | values of register j} and src = {gates of j + 8 | gates of j} by lane half — lanes 0..15 finish the row of register j + 8, lanes
// 16..31 the row of register j, 16 consecutive outputs each (32-byte row segments).  Same arithmetic per output as epi_geglu.
template <int RB, int CB>
__device__ __forceinline__ void epi_geglu16(const float16_t (&acc)[RB][CB], const G16Args& g, int64_t row0, int col0, int wr, int wc, int lane) {
    const int hi = lane >> 5, l16 = lane & 15, up = (lane >> 4) & 1;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int oc0 = (col0 / 32 + wc * CB + cb) * 16;  // first output of this column block
        if (oc0 >= g.geglu_inner) continue;
        const float bx = g.ep.bias ? g.ep.bias[oc0 + l16] : 0.f, bg = g.ep.bias ? g.ep.bias[g.geglu_inner + oc0 + l16] : 0.f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int64_t base_row = row0 + wr * (RB * 32) + rb * 32;
            if (base_row >= g.R) continue;
            const int nv      = (int)(g.R - base_row < 32 ? g.R - base_row : 32);
            _Float16* ub      = g.dst16 + base_row * g.ldd16 + oc0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[rb][cb][j + 8]), __float_as_uint(acc[rb][cb][j]), false, false);
                const float xv = __uint_as_float(sw[0]) * g.ep.scale + bx, gv = __uint_as_float(sw[1]) * g.ep.scale + bg;
                const int r  = up ? j : j + 8;                      // the register this lane finishes
                const int ro = (r & 3) + 8 * (r >> 2) + 4 * hi;     // its row inside the block
                if (ro < nv) ub[(int64_t)ro * g.ldd16 + l16] = (_Float16)(xv * act_apply<UN_GELU>(gv));
            }
        }
    }
}

// linear (D[row][col]): register r holds row ro(r) + 4*hi of the block, lanes run along columns.  Fast variants need a full row tile.
template <int MODE, int RB, int CB>
__device__ __forceinline__ void epi_linear(const float16_t (&acc)[RB][CB], const G16Args& g, int64_t row0, int col0, int wr, int wc, int lane) {
    const int hi = lane >> 5, lc = lane & 31;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t base_row = row0 + wr * (RB * 32) + rb * 32;  // wave-uniform
        if ((MODE == EPI_GENERIC || MODE == EPI_F16 || MODE == EPI_F16_GELU || MODE == EPI_F16_RES || MODE == EPI_F32_GATE) && base_row >= g.R) continue;
        uint32_t hm_n0 = 0, hm_l0 = 0;
        if (MODE == EPI_HM_F32 || MODE == EPI_HM_F16) {
            if (base_row >= g.R) continue;
            hm_n0 = (uint32_t)base_row / (uint32_t)g.hm_L;
            hm_l0 = (uint32_t)base_row - hm_n0 * (uint32_t)g.hm_L;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int cblk = col0 + (wc * CB + cb) * 32;
            const int col  = cblk + lc;
            if (col >= g.C) continue;
            const float bias = g.ep.bias ? g.ep.bias[col] : 0.f;
            if (MODE == EPI_F32 || MODE == EPI_F32_RES) {
                float* ub         = g.dst + base_row * g.ldd + cblk;
                const float* ur   = MODE == EPI_F32_RES ? g.ep.residual + base_row * g.ldd + cblk : nullptr;
                const uint32_t lb = (uint32_t)(lc + 4 * hi * (int)g.ldd) * 4u;
                // all residual loads are issued before the first store (dst may BE the residual: the loads cannot be hoisted by the compiler)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float rv[8];
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r) rv[r - r0] = MODE == EPI_F32_RES ? ld_u(ur + (int64_t)((r & 3) + 8 * (r >> 2)) * g.ldd, lb) : 0.f;
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r)
                        st_u(ub + (int64_t)((r & 3) + 8 * (r >> 2)) * g.ldd, lb, acc[rb][cb][r] * g.ep.scale + bias + rv[r - r0]);
                }
            } else if (MODE == EPI_F16 || MODE == EPI_F16_GELU) {
                _Float16* ub      = g.dst16 + base_row * g.ldd16 + cblk;
                const uint32_t lb = (uint32_t)(lc + 4 * hi * (int)g.ldd16) * 2u;
                const int nvl     = (int)(g.R - base_row < 32 ? g.R - base_row : 32) - 4 * hi;  // ragged last row tile: masked per register
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (ro >= nvl) continue;
                    float v = acc[rb][cb][r] * g.ep.scale + bias;
                    if (MODE == EPI_F16_GELU) v = act_apply<UN_GELU>(v);
                    st_u(ub + (int64_t)ro * g.ldd16, lb, (_Float16)v);
                }
            } else if (MODE == EPI_F16_RES) {
                // f16 rows of acc + bias + residual (f32 rows, ldd): the result is read only as a GEMM / 1x1-conv operand image
                _Float16* ub      = g.dst16 + base_row * g.ldd16 + cblk;
                const float* ur   = g.ep.residual + base_row * g.ldd + cblk;
                const uint32_t lb = (uint32_t)(lc + 4 * hi * (int)g.ldd16) * 2u, lr = (uint32_t)(lc + 4 * hi * (int)g.ldd) * 4u;
                const int nvl     = (int)(g.R - base_row < 32 ? g.R - base_row : 32) - 4 * hi;
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float rv[8];
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        rv[r - r0]   = ro < nvl ? ld_u(ur + (int64_t)ro * g.ldd, lr) : 0.f;
                    }
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        if (ro >= nvl) continue;
                        st_u(ub + (int64_t)ro * g.ldd16, lb, (_Float16)(acc[rb][cb][r] * g.ep.scale + bias + rv[r - r0]));
                    }
                }
            } else if (MODE == EPI_F32_GATE) {
                // x + (acc + bias) * gate[image][col]: the 32 rows of a block cross at most one image boundary (gate_L >= 32)
                float* ub         = g.dst + base_row * g.ldd + cblk;
                const float* ur   = g.ep.residual + base_row * g.ldd + cblk;
                const uint32_t lb = (uint32_t)(lc + 4 * hi * (int)g.ldd) * 4u;
                const uint32_t n0 = (uint32_t)base_row / (uint32_t)g.ep.gate_L, l0 = (uint32_t)base_row - n0 * (uint32_t)g.ep.gate_L;
                const int wrap_at = g.ep.gate_L - (int)l0 - 4 * hi;
                const int nvl     = (int)(g.R - base_row < 32 ? g.R - base_row : 32) - 4 * hi;
                const float g0 = g.ep.gate[(int64_t)n0 * g.C + col];
                const float g1 = (int64_t)(n0 + 1) * g.ep.gate_L < g.R ? g.ep.gate[(int64_t)(n0 + 1) * g.C + col] : 0.f;
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float rv[8];
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        rv[r - r0]   = ro < nvl ? ld_u(ur + (int64_t)ro * g.ldd, lb) : 0.f;
                    }
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        if (ro >= nvl) continue;
                        st_u(ub + (int64_t)ro * g.ldd, lb, (acc[rb][cb][r] * g.ep.scale + bias) * (ro >= wrap_at ? g1 : g0) + rv[r - r0]);
                    }
                }
            } else if (MODE == EPI_HM_F32 || MODE == EPI_HM_F16) {
                // attention operand layout [d, L, H, N] — what CONT(permute(0,2,1,3)) (+CPY f16) of the projection would hold:
                // element (row = n*L + l, col = h*d + dd) -> n*(H*L*d) + h*(L*d) + l*d + dd.  L >= 32: the 32 rows of a block cross at
                // most ONE image boundary; rows past it move by (H-1)*L*d elements.  Ragged last row tiles are masked per register.
                const int64_t Ld   = (int64_t)g.hm_L * g.hm_d;
                const int64_t ub   = (int64_t)hm_n0 * Ld * g.hm_H + (int64_t)hm_l0 * g.hm_d;  // uniform
                const uint32_t h   = (uint32_t)col / (uint32_t)g.hm_d, dd = (uint32_t)col - h * (uint32_t)g.hm_d;
                const uint32_t le  = h * (uint32_t)Ld + dd + 4u * hi * (uint32_t)g.hm_d;  // per-lane elements (< one image of the operand)
                const uint32_t adj = (uint32_t)(Ld * (g.hm_H - 1));
                const int wrap_at  = g.hm_L - (int)hm_l0 - 4 * hi;  // register rows ro >= wrap_at belong to the next image
                const int nvl      = (int)(g.R - base_row < 32 ? g.R - base_row : 32) - 4 * hi;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (ro >= nvl) continue;
                    const int64_t uo  = ub + (int64_t)ro * g.hm_d;
                    const uint32_t lo = le + (ro >= wrap_at ? adj : 0u);
                    const float v     = acc[rb][cb][r] * g.ep.scale + bias;
                    if (MODE == EPI_HM_F32)
                        st_u(g.dst + uo, lo * 4u, v);
                    else
                        st_u(g.dst16 + uo, lo * 2u, (_Float16)v);
                }
            } else {
                // everything else (ragged last row tile, f32+f16 double output); head-major stores never come here (L >= 32 is a
                // launch precondition and their variant masks ragged rows itself)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = base_row + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row >= g.R) continue;
                    float v = acc[rb][cb][r] * g.ep.scale + bias;
                    if (g.ep.residual) v += g.ep.residual[row * g.ldd + col];
                    if (g.dst) g.dst[row * g.ldd + col] = v;
                    if (g.dst16) g.dst16[row * g.ldd16 + col] = (_Float16)v;
                }
            }
        }
    }
}

// the Linear epilogue of one workgroup tile: ONE compact variant, chosen from launch- / workgroup-uniform conditions
template <int BM, int RB, int CB>
__device__ __forceinline__ void epi_dispatch_linear(const float16_t (&acc)[RB][CB], const G16Args& g, int64_t row0, int col0, int wr, int wc, int lane) {
    const bool full = row0 + BM <= g.R;
    if (g.geglu_inner > 0 && g.geglu16) {
        epi_geglu16(acc, g, row0, col0, wr, wc, lane);
    } else if (CB % 2 == 0 && g.geglu_inner > 0) {
        if constexpr (CB % 2 == 0) epi_geglu(acc, g, row0, col0, wr, wc, lane);
    } else if (g.hm_d > 0 && g.hm_L >= 32 && !g.ep.residual && (g.dst != nullptr) != (g.dst16 != nullptr)) {
        if (g.dst16)
            epi_linear<EPI_HM_F16>(acc, g, row0, col0, wr, wc, lane);
        else
            epi_linear<EPI_HM_F32>(acc, g, row0, col0, wr, wc, lane);
    } else if (g.ep.gate) {
        epi_linear<EPI_F32_GATE>(acc, g, row0, col0, wr, wc, lane);
    } else if (full && g.hm_d == 0 && g.dst && !g.dst16) {
        if (g.ep.residual)
            epi_linear<EPI_F32_RES>(acc, g, row0, col0, wr, wc, lane);
        else
            epi_linear<EPI_F32>(acc, g, row0, col0, wr, wc, lane);
    } else if (g.hm_d == 0 && g.dst16 && !g.dst && g.ep.residual && !g.ep.gelu) {
        epi_linear<EPI_F16_RES>(acc, g, row0, col0, wr, wc, lane);
    } else if (g.hm_d == 0 && g.dst16 && !g.dst && !g.ep.residual) {
        if (g.ep.gelu)
            epi_linear<EPI_F16_GELU>(acc, g, row0, col0, wr, wc, lane);
        else
            epi_linear<EPI_F16>(acc, g, row0, col0, wr, wc, lane);
    } else {
        epi_linear<EPI_GENERIC>(acc, g, row0, col0, wr, wc, lane);
    }
}

// ---- SWP (option "gemm16_swp"; default off.  Round 4 ran it: correct — 98 Linear / GEGLU / model tests with it forced on — and 1.4 % SLOWER per SD1.5 step,
// 23.24 -> 23.56 ms, profiles/r05a_ab_gemm16_swp_rejected.txt: an access instruction that touches 32 rows x 32 bytes costs more than four that touch 2 rows x 128
// bytes, and narrow stores are not what bounds these epilogues, profiles/r05b_store_width_probe.txt) ----
// The Linear kernels of the big-token tiles instantiated with the MFMA operands swapped (as the conv path does): a 32x32 accumulator block then
// holds D^T — lane = ROW (lane & 31) of the block, register r = COLUMN (r&3) + 8*(r>>2) + 4*hi — so a lane owns four CONSECUTIVE output features
// of one token per register quad: bias, residual and result move as 16-byte (f32) / 8-byte (f16) accesses, one per four elements, and a ragged
// last row tile is a per-lane predicate.  The D[row][col] layout needs one 4-byte (2-byte) access per element: at K = 320 the 160-element epilogue
// of a wave issues more instructions than its 100-MFMA main loop.  What it costs: an access instruction touches 32 rows x 32
// bytes instead of 2 rows x 128 bytes — and that is what the GPU run charged for (above).
// Serves: f32 (+bias, +residual), f16 rows (+GELU), GEGLU.  Everything else stays on the D[row][col] kernels (host-side choice, g16_swp_ok).
enum { SWP_F32 = 0, SWP_F16 = 1, SWP_GEGLU = 2 };
template <int MODE, int RB, int CB>
__device__ __forceinline__ void epi_linear_swp(const float16_t (&acc)[RB][CB], const G16Args& g, int64_t row0, int col0, int wr, int wc, int lane) {
    const int hi = lane >> 5, lr = lane & 31;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t row = row0 + wr * (RB * 32) + rb * 32 + lr;  // this lane's token
        if (row >= g.R) continue;
        if constexpr (MODE == SWP_GEGLU) {
            static_assert(MODE != SWP_GEGLU || CB % 2 == 0, "GEGLU pairing needs an even number of column blocks per wave");
#pragma unroll
            for (int p = 0; p < CB / 2; ++p) {
                const int gb  = col0 / 32 + wc * CB + 2 * p;                  // global 32-column block of the value half (see epi_geglu)
                const int oc0 = ((gb >> 2) * 2 + ((gb & 3) >> 1)) * 32;      // first output column of this pair
                if (oc0 >= g.geglu_inner) continue;
                _Float16* orow = g.dst16 + row * g.ldd16 + oc0 + 4 * hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f), bg = bx;
                    if (g.ep.bias) {
                        bx = *(const float4*)(g.ep.bias + oc0 + 8 * q + 4 * hi);
                        bg = *(const float4*)(g.ep.bias + g.geglu_inner + oc0 + 8 * q + 4 * hi);
                    }
                    const float bxs[4] = {bx.x, bx.y, bx.z, bx.w}, bgs[4] = {bg.x, bg.y, bg.z, bg.w};
                    half4_t h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = acc[rb][2 * p][4 * q + j] * g.ep.scale + bxs[j], gv = acc[rb][2 * p + 1][4 * q + j] * g.ep.scale + bgs[j];
                        h[j]           = (_Float16)(xv * act_apply<UN_GELU>(gv));
                    }
                    *(half4_t*)(orow + 8 * q) = h;
                }
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int cblk = col0 + (wc * CB + cb) * 32;
                if (cblk >= g.C) continue;  // whole blocks only: C % 32 == 0 is a launch precondition
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = cblk + 8 * q + 4 * hi;
                    float4 b    = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g.ep.bias) b = *(const float4*)(g.ep.bias + c);
                    const float bs[4] = {b.x, b.y, b.z, b.w};
                    if constexpr (MODE == SWP_F32) {
                        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (g.ep.residual) rv = *(const float4*)(g.ep.residual + row * g.ldd + c);  // the same lane stores these four elements below
                        float4 o;
                        o.x = acc[rb][cb][4 * q] * g.ep.scale + bs[0] + rv.x;
                        o.y = acc[rb][cb][4 * q + 1] * g.ep.scale + bs[1] + rv.y;
                        o.z = acc[rb][cb][4 * q + 2] * g.ep.scale + bs[2] + rv.z;
                        o.w = acc[rb][cb][4 * q + 3] * g.ep.scale + bs[3] + rv.w;
                        *(float4*)(g.dst + row * g.ldd + c) = o;
                    } else {
                        half4_t h;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = acc[rb][cb][4 * q + j] * g.ep.scale + bs[j];
                            if (g.ep.gelu) v = act_apply<UN_GELU>(v);
                            h[j] = (_Float16)v;
                        }
                        *(half4_t*)(g.dst16 + row * g.ldd16 + c) = h;
                    }
                }
            }
        }
    }
}
template <int RB, int CB>
__device__ __forceinline__ void epi_dispatch_linear_swp(const float16_t (&acc)[RB][CB], const G16Args& g, int64_t row0, int col0, int wr, int wc, int lane) {
    if (g.geglu_inner > 0) {
        if constexpr (CB % 2 == 0) epi_linear_swp<SWP_GEGLU>(acc, g, row0, col0, wr, wc, lane);
    } else if (g.dst16) {
        epi_linear_swp<SWP_F16>(acc, g, row0, col0, wr, wc, lane);
    } else {
        epi_linear_swp<SWP_F32>(acc, g, row0, col0, wr, wc, lane);
    }
}

// conv (D[oc][pos]): register r holds output channel ro(r) + 4*hi of the block, lanes run along output positions.
// MODE 0: bias only, 1: + residual, 2: generic (ragged channel block)
template <int MODE, int RB, int CB>
__device__ __forceinline__ void epi_conv(const float16_t (&acc)[RB][CB], const G16Args& g, int64_t row0, int col0, int wr, int wc, int lane) {
    const int hi = lane >> 5, lc = lane & 31;
    const uint32_t ohow = (uint32_t)g.OHOW;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t pos_base = row0 + wr * (RB * 32) + rb * 32;  // wave-uniform
        if (pos_base >= g.R) continue;
        const uint32_t img0 = (uint32_t)pos_base / ohow, p0 = (uint32_t)pos_base - img0 * ohow;
        uint32_t pl = p0 + lc, dimg = 0;
        while (pl >= ohow) {  // a 32-position block may straddle images when OH*OW % 32 != 0
            pl -= ohow;
            ++dimg;
        }
        if (pos_base + lc >= g.R) continue;
        const uint32_t le = dimg * (uint32_t)g.C * ohow + pl + 4u * hi * ohow;  // per-lane elements relative to (img0, cblk)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int cblk = col0 + (wc * CB + cb) * 32;
            if (cblk >= g.C) continue;
            const int64_t ub = ((int64_t)img0 * g.C + cblk) * g.OHOW;  // uniform
            const float* pb  = g.ep.bias ? g.ep.bias + cblk : nullptr;
            const uint32_t cld = g.ep.chan_ld ? (uint32_t)g.ep.chan_ld : (uint32_t)g.C;  // floats between the images of chan_add
            const float* pc  = g.ep.chan_add ? g.ep.chan_add + (int64_t)img0 * cld + cblk : nullptr;  // + dimg * cld per lane
            // loads first (bias, residual), then the stores: dst may BE the residual, so the compiler cannot batch them itself
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float bv[8], rv[8];
#pragma unroll
                for (int r = r0; r < r0 + 8; ++r) {
                    const int ro  = (r & 3) + 8 * (r >> 2);
                    const bool ok = MODE != 2 || cblk + ro + 4 * hi < g.C;
                    bv[r - r0]    = (pb && ok) ? ld_u(pb + ro, 16u * hi) : 0.f;
                    if (pc && ok) bv[r - r0] += ld_u(pc + ro, (4u * hi + dimg * cld) * 4u);
                    rv[r - r0]    = ((MODE == 1 || (MODE == 2 && g.ep.residual)) && ok) ? ld_u(g.ep.residual + ub + (int64_t)ro * g.OHOW, le * 4u) : 0.f;
                }
#pragma unroll
                for (int r = r0; r < r0 + 8; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (MODE == 2 && cblk + ro + 4 * hi >= g.C) continue;
                    st_u(g.dst + ub + (int64_t)ro * g.OHOW, le * 4u, acc[rb][cb][r] * g.ep.scale + bv[r - r0] + rv[r - r0]);
                }
            }
        }
    }
}

}  // namespace mi355x
