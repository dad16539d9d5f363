// kernels.h — host-callable launchers of the hand-written gfx950 kernels.  Every launcher enqueues on the
// given HIP stream and returns immediately.  Shapes use ggml's ne order (ne0 contiguous).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mi355x {

// 4-D strided view of a tensor (element strides are in BYTES like ggml's nb[])
struct View4 {
    const void* data;
    int64_t ne[4];
    int64_t nb[4];
    int type;  // ggml_type
};

enum BinOp { BIN_ADD = 0, BIN_SUB = 1, BIN_MUL = 2, BIN_DIV = 3 };
enum UnOp { UN_SILU = 0, UN_GELU = 1, UN_GELU_QUICK = 2, UN_SIGMOID = 3, UN_TANH = 4, UN_RELU = 5, UN_NEG = 6, UN_EXP = 7 };

// ---- elementwise.hip --------------------------------------------------------------------------------
// dst = a (op) broadcast(b); a/dst f32 with arbitrary strides, b f32 broadcast by modulo on every dim
void launch_binary(hipStream_t s, BinOp op, void* dst, const int64_t dnb[4], const View4& a, const View4& b);
void launch_unary(hipStream_t s, UnOp op, float* dst, const float* src, int64_t n);
void launch_scale(hipStream_t s, float* dst, const float* src, int64_t n, float scale, float bias);
// generic strided copy with conversion between f32/f16/bf16 (logical element order preserved)
void launch_copy(hipStream_t s, const View4& dst, const View4& src);
void launch_concat(hipStream_t s, const View4& dst, const View4& a, const View4& b, int dim);
void launch_repeat(hipStream_t s, const View4& dst, const View4& src);
// a [d*H, La, N], b [d*H, Lb, N] f32 contiguous -> out [d, La+Lb, H, N] (f32 or f16): token concat + head-major permute (+ cast) in one pass
void launch_concat_heads(hipStream_t s, void* out, bool out_f16, const float* a, const float* b, int64_t d, int64_t H, int64_t La, int64_t Lb, int64_t N);
// one of q / k / v of a joint attention: both streams' columns of their fused qkv projections (row strides xsa / xsb floats) -> optional per-head
// RMSNorm * w -> token concat (a first) -> head-major [d, La + Lb, H, N], f32 or f16 (elementwise.hip k_joint_heads); d = 64 or 128
bool joint_heads_supported(int64_t d);
void launch_joint_heads(hipStream_t s, void* out, bool out_f16, const float* a, int64_t xsa, const float* wa, const float* b, int64_t xsb, const float* wb, float eps,
                        int64_t d, int64_t H, int64_t La, int64_t Lb, int64_t N, const float* pe = nullptr);  // pe: rotary table [2,2,d/2,La+Lb] applied after the norm (FLUX)
// interleaved rotary embedding: x [d, H, L, N] (d contiguous, other dims strided), pe [2,2,d/2,L] -> out [d, L, H*N] contiguous
void launch_rope_pairs(hipStream_t s, float* out, const View4& x, const float* pe);
void launch_upscale_nearest(hipStream_t s, const View4& dst, const View4& src);
void launch_pad(hipStream_t s, const View4& dst, const View4& src, const int32_t pads[8]);
// GET_ROWS: ids i32 [ne0, ne1, ne2] (byte strides), table f32/f16/bf16/q8_0/q4_0 [nc, rows, ne1, ne2] -> dst f32 [nc, ne0, ne1, ne2]
void launch_get_rows(hipStream_t s, float* dst, const int64_t dnb[4], const View4& table, const View4& ids);
void launch_timestep_embedding(hipStream_t s, float* dst, const float* t, int n, int dim, int max_period, int64_t dst_row_stride);
// GEGLU: dst[t][i] = x[t][i] * gelu(x[t][inner + i]); x row stride given in floats
void launch_geglu(hipStream_t s, float* dst, const float* x, int64_t tokens, int64_t inner, int64_t x_stride);

// ---- norm.hip ---------------------------------------------------------------------------------------
// GROUP_NORM over [W*H, C, N] contiguous f32; optional fused affine (w,b per channel) and SiLU
void launch_group_norm(hipStream_t s, float* dst, const float* x, int64_t hw, int64_t C, int64_t N, int groups, float eps,
                       const float* w, const float* b, bool silu);
// NORM / RMS_NORM over rows of ne0 contiguous f32 (row strides in floats); optional fused affine
void launch_layer_norm(hipStream_t s, float* dst, const float* x, int64_t ne0, int64_t nrows, int64_t x_stride, int64_t d_stride,
                       float eps, const float* w, const float* b, bool rms);
// rows addressed by (i1, i2, i3) with byte strides xnb / dnb (strided views: per-head slices of a fused qkv projection)
void launch_layer_norm_4d(hipStream_t s, float* dst, const float* x, const int64_t ne[4], const int64_t xnb[4], const int64_t dnb[4], float eps, const float* w,
                          const float* b, bool rms);
void launch_soft_max(hipStream_t s, float* dst, const float* x, int64_t ncols, int64_t nrows, float scale, const View4* mask,
                     int64_t rows_per_mat);
// softmax over contiguous f32 rows (ncols % 4 == 0) -> f16 operand image [nrows][ncols rounded up to 64]
void launch_soft_max_rows_f16(hipStream_t s, void* dst16, const float* x, int64_t ncols, int64_t nrows);

// ---- gemm_generic.hip: any-operand matmul (exact f32 MFMA), ggml MUL_MAT semantics ------------------
// dst[m + n*ldd] = sum_k A[m][k]*B[n][k];  A type in {f32,f16,bf16,q8_0,q4_0} rows K-contiguous, B f32/f16
void launch_mul_mat_generic(hipStream_t s, float* dst, const int64_t dne[4], const int64_t dnb[4], const View4& a, const View4& b);
void launch_im2col_f16(hipStream_t s, void* dst, int dst_type, const View4& x, int64_t KW, int64_t KH, int64_t OW, int64_t OH,
                       int s0, int s1, int p0, int p1, int d0, int d1);

// ---- wgemm.hip: static-weight GEMMs on f16 MFMA (32x32x16), weights pre-swizzled into fragment order ----
// pre-swizzle: rows R (padded to 32) x K (padded to 16) -> [R/32][K/16][64 lanes][8 halfs]
size_t wswz_bytes(int64_t R, int64_t K);
// geglu_inner > 0: rows permuted so that every wave of a gemm16 column tile owns matching value / gate blocks (GEGLU epilogue)
void launch_wswz_linear(hipStream_t s, void* dst, const void* src, int src_type, int64_t K, int64_t R, int64_t src_row_bytes, int64_t geglu_inner = 0);
// conv weight [KW,KH,IC,OC] f16 -> rows OC, K index = tap*ICp + ic (ICp = IC padded to 32)
void launch_wswz_conv(hipStream_t s, void* dst, const void* src, int64_t KW, int64_t KH, int64_t IC, int64_t OC, int kblk = 0);
// kblk: K ordered (kblk-channel block, tap, channel): 64 for the per-tap gather kernels of gemm16.hip, 32 for the LDS-window kernel of conv3w.hip;
// 0: (tap, channel)

struct Epilogue {
    const float* bias     = nullptr;  // per output feature / channel
    const float* residual = nullptr;  // same layout as dst (added after bias)
    const float* chan_add = nullptr;  // conv only: per (oc, n) value added (time-embedding broadcast), [OC, N]
    int64_t chan_ld       = 0;        // floats between the images of chan_add (0 = OC: the graph tensor; > OC: a column range of a grouped projection's output)
    float scale           = 1.0f;     // applied to the accumulator before bias
    int act               = -1;       // UnOp applied last, or -1
    // conv only, split-K launches only: the GroupNorm that reads this output next (planner look-ahead) — the slab reduce pass then also computes the
    // group statistics of the values it writes and stores the per-(image, channel) affine y = x * gn_scale + gn_shift (k_splitk_reduce_gn)
    float* gn_scale       = nullptr;
    float* gn_shift       = nullptr;
    const float* gn_w     = nullptr;
    const float* gn_b     = nullptr;
    int gn_groups         = 0;
    float gn_eps          = 0.f;
    // Linear only, split-K launches with the slab reduce pass only: the LayerNorm (-> MUL w -> ADD b) that reads this output next and feeds only
    // weight GEMMs (planner look-ahead, plan_linear) — the reduce pass keeps each finished row in registers and also writes that LayerNorm's
    // f16 operand image (k_splitk_reduce_ln): no second read of the tensor, no second launch
    void* ln_dst16        = nullptr;  // [rows][rup64(M)] halfs
    const float* ln_w     = nullptr;
    const float* ln_b     = nullptr;
    float ln_eps          = 0.f;
    // gemm16 linear only (DiT blocks):
    const float* gate     = nullptr;  // [images][M]: dst = (acc*scale + bias) * gate[row / gate_L][col] + residual
    int gate_L            = 0;        // rows per image
    int gelu              = 0;        // f16-only output: tanh-GELU applied before rounding
    // gemm16 linear only, plain f32 output launches: the column tiles at or beyond split_col do not write the f32 output — they store
    // gelu(acc * scale + bias) as f16 into another tensor: element (row, col) at split_dst16[row * split_ldd16 + (col - split_col)].  FLUX single
    // block (flux.hpp:594-700): linear1 = [q k v | mlp]; the mlp columns are read only as gelu(mlp) inside linear2's operand image.  split_col must be
    // a multiple of 256 and of the launch's column tile.
    int64_t split_col     = 0;
    void* split_dst16     = nullptr;
    int64_t split_ldd16   = 0;
    // gemm16 linear only: the weight pointer(s) handed to the launch are NOT f16 images but the RAW GGUF rows of a q8_0 (qtype 8) / q4_0 (qtype 2) tensor,
    // qrow_bytes apart; the blocks are dequantised inside the GEMM's main loop (k_gemm16<..., QT>).  Only for launches gemm16_qinloop_supported accepts.
    int qtype             = 0;
    int64_t qrow_bytes    = 0;
    // gemm16 / qgemm16 linear only: the f16 operand image holds the rows in RUNS — row r of the Linear is image row (r / a_run_L) * a_run_S + r % a_run_L (the pointer
    // handed to the launch addresses row 0).  A token slice [C, L, N] of an attention output whose f16 image the flash kernel wrote for all (Lq > L) tokens of each of
    // the N images (MMDiT block_mixing, mmdit.hpp:651-667): a_run_L = L, a_run_S = Lq.  0 = rows are consecutive.
    int64_t a_run_L       = 0;
    int64_t a_run_S       = 0;
};
// ---- gemm16.hip: second-generation contraction, both operands f16 via LDS-DMA -------------------------------
// a16: f16 row-major [rows][lda] (K contiguous, padded to 64); output f32 [rows][ldd] and/or f16 [rows][ldd16]
void gemm16_init();
// a Linear of this shape over raw q8_0 (wtype 8) / q4_0 (wtype 2) rows runs on the pipelined 256 x 256 tile with the blocks dequantised in the main loop:
// no f16 weight image, resident or rebuilt (mul: sibling weights in one launch; split: the launch's K slices, <= 1 = none)
bool gemm16_qinloop_supported(int wtype, int64_t rows, int64_t M, int64_t K, int mul, int split);
void gemm16_set_qinloop_min_rows(int v);
void gemm16_set_t192p(int v);  // 0: the per-shape choice never takes the pipelined 256 x 192 tile
void gemm16_set_tap_major(int v);  // A/B: conv K order (tap, channel block) instead of (channel block, tap)
int gemm16_tap_major();
void gemm16_set_splitk_target(int v);
void gemm16_set_splitk_mid(int v);  // 1: two K slices for launches of 193..384 workgroups with >= 128 K tiles (experiment, default 0)
void gemm16_set_tile(int t);     // -1: per-shape choice; 0..5: force T128 / T256 / T256W / T160 / T160N / T320 (A/B measurements)
#ifdef MI355X_EXPERIMENTS
void gemm16_set_abl(int v);
#endif
void gemm16_set_t320(int v);
void gemm16_set_t256p_pad(int v);   // option "t256p_pad" (1): 256 x 256 pipelined tile for Linear widths that are multiples of 128 only (last column tile half empty)
void gemm16_set_conv_wmajor(int v);  // option "conv_wmajor" (1): weight-major workgroup order for convs whose weight image is >= 2x their input image
void gemm16_set_ln16_rows(int v);    // option "ln16_rows" (4): rows per wave of the LayerNorm -> f16 operand image kernel for rows of <= 1280 values
void gemm16_set_tail_split(int v);  // option "tail_split" (1): row-split launches (whole rounds of 256 x 256 tiles + the remaining rows on small tiles)
void gemm16_set_bn64(int v);     // 0: never choose the pipelined 256x320 tile
void gemm16_set_variant(int v);  // 0: BK64x2 stages, 1: BK32x3 stages (default), 2: BK64x3 stages
// hm_d > 0: head-major store — element (row = n*hm_L + l, col = h*hm_d + dd) goes to ((n*hm_H + h)*hm_L + l)*hm_d + dd of dst (f32) / dst16 (f16)
void launch_gemm16_linear(hipStream_t s, float* dst, void* dst16, int64_t ldd16, const void* a16, int64_t lda, const void* wswz, int64_t rows,
                          int64_t K, int64_t M, int64_t ldd, const Epilogue& ep, int hm_d = 0, int hm_H = 0, int hm_L = 0, float* splitk_ws = nullptr, int* splitk_cnt = nullptr, int splitk_S = 0);
// n (2..16) sibling Linears over the same operand image in one launch; per weight: image, f32 and / or f16 destination (head-major when hm_d > 0), bias
void launch_gemm16_linear_multi(hipStream_t s, int n, float* const* dst, void* const* dst16, const void* a16, int64_t lda, const void* const* wswz, int64_t rows,
                               int64_t K, int64_t M, const float* const* bias, float scale, int hm_d, int hm_H, int hm_L);
// FF1 + GEGLU in one kernel (block.hpp:193-210): wswz built with geglu_inner = M/2; dst16[t][c] = (y[t][c] + b[c]) * gelu(y[t][inner + c] + b[inner + c]),
// f16 row-major with row stride inner (inner % 64 == 0) — the operand image of the FF2 GEMM.  The [tokens][2*inner] f32 tensor is never written.
void launch_gemm16_linear_geglu(hipStream_t s, void* dst16, const void* a16, int64_t lda, const void* wswz_geglu, int64_t rows, int64_t K, int64_t M,
                                const float* bias, float* splitk_ws = nullptr, int* splitk_cnt = nullptr, int splitk_S = 0, int geglu_mode = 1);
// layout of the weight image a GEGLU FF1 of this shape needs: 1 = 128-column value / gate pairing, 2 = 16-column interleave (launch_wswz_linear geglu_inner < 0)
int gemm16_geglu_mode(int64_t rows, int64_t M, int64_t K);
void gemm16_set_bn64_max(int v);  // option "bn64_max_tiles" (128)
void gemm16_set_geglu16(int v);  // option "geglu16" (1)
// split-K factor the launchers will use when given a workspace of factor * rows * M floats (1 = no split)
int gemm16_split_k(int64_t rows, int64_t M, int64_t K, bool conv);
bool splitk_reduce_gn_supported(int64_t hw, int64_t C, int64_t N, int groups);
bool splitk_reduce_ln_supported(int64_t rows, int64_t M);  // the slab reduce of a split Linear can also write the next LayerNorm's f16 operand image  // the slab reduce of a split conv can also produce the next GroupNorm's statistics
void gemm16_set_t320_linear_max_split(int v);  // option "t320_linear_max_split" (4): most K slices a Linear takes on the 256x320 tile
// the split a launch of this shape should take: S slices; inkernel = combined by the last-arriving workgroup of every output tile (the launcher
// then needs `tiles` zeroed int counters and ws_bytes of slab space, and applies the full epilogue itself), else slabs + k_splitk_reduce (only
// for plain f32 outputs: plain_out).  S = 1: no split.
struct G16SplitPlan {
    int S;
    bool inkernel;
    size_t ws_bytes;
    int tiles;
};
G16SplitPlan gemm16_split_plan(int64_t rows, int64_t M, int64_t K, bool conv, bool plain_out, bool geglu = false);  // S < 0: stream-K with -S persistent workgroups (Linear only)
void gemm16_set_splitk_inkernel(int v);
void gemm16_set_splitk_in_target(int v);
// x16: f16 NHWC [N][H][W][ICp]; dst f32 NCHW [OW,OH,OC,N]
void launch_gemm16_conv(hipStream_t s, float* dst, const void* x16_nhwc, const void* wswz, int64_t W, int64_t H, int64_t IC, int64_t N, int64_t OC,
                        int ksize, int stride, int pad, bool upscale2x, const Epilogue& ep, float* splitk_ws = nullptr, int* splitk_cnt = nullptr, int splitk_S = 0);
// ---- conv3w.hip: 3x3 / stride-1 conv with the input window resident in LDS (weights in the kblk = 32 image)
// K slices the window kernel would run this shape with (>= 1), or 0: the shape stays on launch_gemm16_conv
int conv3w_plan(int64_t W, int64_t H, int64_t IC, int64_t N, int64_t OC, int ksize, int stride, bool upscale2x, int* bn_out = nullptr);
void launch_conv3w(hipStream_t s, float* dst, const void* x16_nhwc, const void* wswz32, int64_t W, int64_t H, int64_t IC, int64_t N, int64_t OC, const Epilogue& ep,
                   float* splitk_ws, int S);
void qgemm16_set_rb(int v);   // option "qgemm16_rb" (3): row blocks per k_qgemm16 tile (1 / 2 / 4 forced; 3 = 64-row tiles for small grids; 0 = by row count only)
void conv3w_set_prio(int v);  // option "conv3w_prio" (3): static wave priority of the second half of a k_conv3w workgroup; 0 = without (A/B)
void conv3w_set(int v);  // option "conv3w"
void conv3w_set_min_blocks(int v);  // option "conv3w_min_blocks"
void conv3w_set_min_blocks_deep(int v);  // option "conv3w_min_blocks_deep"
// producers of f16 operand images (row stride = K rounded up to 64, zero padded)
// L > 0: rows are N runs of L rows, run n starting bs elements after run n-1 (a token slice of a [C, Lfull, N] tensor)
void launch_pack_rows_f16(hipStream_t s, void* dst, const float* x, int64_t R, int64_t K, int64_t xs, int64_t L = 0, int64_t bs = 0);
// f32 rows -> K columns of f16 rows with stride ld (one part of an operand image filled by several producers), optional tanh-GELU; K % 8 == 0
void launch_pack_cols_f16(hipStream_t s, void* dst, int64_t ld, const float* x, int64_t R, int64_t K, int64_t xs, bool gelu);
// mod_L > 0: w, b are per-image [rows / mod_L][ne0] adaLN tables and the affine is norm * (1 + w) + b (DiT modulate)
void launch_layer_norm_f16(hipStream_t s, void* dst, const float* x, int64_t ne0, int64_t nrows, int64_t x_stride, float eps, const float* w,
                           const float* b, bool rms, int64_t mod_L = 0);
void launch_geglu_f16(hipStream_t s, void* dst, const float* x, int64_t tokens, int64_t inner, int64_t x_stride);
// x2 != nullptr (both launchers): the activation is the channel concatenation [x (C1 channels) | x2 (C - C1 channels)] of two NCHW tensors, never materialised
// (UNet skip connections); dst_raw != nullptr: a second NHWC image of the same values without affine / SiLU
bool gn_two_source_supported(const float* x, const float* x2, int64_t hw, int64_t C, int64_t C1, int groups);
// part: optional scratch of N * groups * gn_stats_split(...) * 2 floats — with it, few large (image, group) slabs are shared by several workgroups
void launch_gn_stats(hipStream_t s, float* scale, float* shift, const float* x, int64_t hw, int64_t C, int64_t N, int groups, float eps,
                     const float* w, const float* b, const float* x2 = nullptr, int64_t C1 = 0, float* part = nullptr);
int gn_stats_split(int64_t hw, int64_t C, int64_t N, int groups);
void gemm16_set_gn_split_min(int v);  // option "gn_split_min" (65536 floats): least slab size for it; 0 = never  // workgroups per slab the split form would use (0: not used for this shape)
void launch_nchw_to_nhwc_f16(hipStream_t s, void* dst, const float* x, int64_t hw, int64_t C, int64_t N, const float* scale, const float* shift,
                             int act, const float* x2 = nullptr, int64_t C1 = 0, void* dst_raw = nullptr, float post_mul = 1.f, float* dst_f32 = nullptr);  // act: 0 none, 1 SiLU, 2 ReLU (after the affine); post_mul: after affine / activation (Conv2d scale); dst_f32: the activated values also as f32 NCHW (may be x itself: the in-place unary), single source only

// ---- qgemm.hip: q8_0 / q4_0 Linear with <= 4 activation rows: raw quantised blocks streamed once, in-register dequant ----------------
bool qgemv_supported(int wtype, int64_t rows, int64_t K);
void qgemv_set_max_rows(int v);
size_t qgemv_workspace_bytes(int64_t rows, int64_t K);
// grouped weight-streaming launch (qgemm.hip): members_dev = device table of n_members entries written with qgemv_fill_member / qgemv_member_bytes
void launch_qgemv_group(hipStream_t s, float* dst, int64_t Mtot, const float* x, int64_t xs, int64_t rows, const void* members_dev, int n_members, int wtype, int64_t K,
                        float pre_scale, bool pre_silu);
size_t qgemv_member_bytes();
void qgemv_fill_member(void* host_entry, const void* W, const float* bias, int start);
void launch_qgemv(hipStream_t s, float* dst, int64_t ldd, const float* x, int64_t xs, int64_t rows, const void* wraw, int wtype, int64_t K, int64_t M, void* ws,
                  const Epilogue& ep, float pre_scale, bool pre_silu = false);

// k_fgemv: f16 / f32 weights under <= 16 activation rows (time-embedding MLP, ResBlock embedding projections): one launch, optional SiLU on the
// activation rows (the UNARY node in front of the Linear is not executed), f32 x f32 for f32 weights
bool fgemv_supported(int wtype, int64_t rows, int64_t K);
void fgemv_set_max_rows(int v);
void launch_fgemv(hipStream_t s, float* dst, int64_t ldd, const float* x, int64_t xs, int64_t rows, const void* w, int wtype, int64_t K, int64_t M, const Epilogue& ep,
                  bool pre_silu);
// k_qgemm16: the same raw-block stream on the MFMA units for 3 .. qgemm16_max_rows activation rows.  a16 = the f16 operand image [rows][lda]
// (lda >= K); plain f32 output [rows][M] (+ bias, residual, gate) or, with ep.gelu, the f16 rows dst16 (row stride ldd16).  Split-K (S slices,
// workspace of S * rows * M floats) only for plain outputs.
bool qgemm16_supported(int wtype, int64_t rows, int64_t K, int64_t M);
int qgemm16_split_k(int64_t rows, int64_t K, int64_t M);
// raw q8_0 / q4_0 rows -> the f16 MFMA weight image of k_gemm16 (plain row order), fast enough to run in front of every launch (planner option jit_qimages)
bool wswz_q_supported(int wtype, int64_t K);
void launch_wswz_q(hipStream_t s, void* dst, const void* wraw, int wtype, int64_t K, int64_t R);
void qgemm16_set_max_rows(int v);
void qgemm16_set_pf(int v);  // segments of global loads in flight in k_qgemm16 (1 / 2)
void launch_qgemm16(hipStream_t s, float* dst, void* dst16, int64_t ldd16, const void* a16, int64_t lda, int64_t rows, const void* wraw, int wtype, int64_t K,
                    int64_t M, const Epilogue& ep, float* splitk_ws = nullptr, int splitk_S = 1);
// dst[i] = sum_s ws[s * n + i] + bias[i % C] + residual[i] (gemm16.hip's k_splitk_reduce on a row-major [rows][C] output)
void splitk_reduce_rows(hipStream_t s, float* dst, const float* ws, int S, int64_t n, const float* bias, int64_t C, const float* residual);

// ---- flash_attn.hip ---------------------------------------------------------------------------------
// q [D,Lq,HN] (f32, strides in bytes), k [D,Lk,HN], v [DV,Lk,HN] (f16 or f32; v may be a transposed view,
// any nb[0]) -> dst f32 written with strides
// (dst_nb_d=4 implied) dst element (d, q, hn) at dst + q*dst_nb_q + hn*dst_nb_h
bool flash_attn_supported(int64_t D, int64_t DV);
// output: f32 element (d, q, hn) at dst + q*dst_nb_q + (hn % H)*dst_nb_h + (hn / H)*dst_nb_n   (H = 0: flat head index),
// and/or f16 operand image dst16[(n*Lq + q)*ld16 + h*D + d] (the packed [tok][C] input of the to_out projection)
struct FlashOut {
    float* dst       = nullptr;
    int64_t nb_q = 0, nb_h = 0, nb_n = 0;
    int H            = 0;
    void* dst16      = nullptr;
    int64_t ld16     = 0;
};
#ifdef MI355X_EXPERIMENTS
void flash_attn_set_ablate(int v);
#endif
void flash_attn_set_grid(int v);   // option "flash_grid"
void flash_attn_set_qb2(int v);    // option "flash_qb2": 1 = two query blocks per wave for the d = 40 max-slot launches (default), 2 = for every d <= 48 launch, 0 = off
void flash_attn_set_pp(int v);     // option "flash_pp": the 8-wave ping-pong kernel (default 0: measured slower; 1 = d in (64, 96], 2 = wherever legal)
void flash_attn_set_vpf(int v);    // option "flash_vpf": bit mask of head-dim classes (1: d <= 48, 2: <= 64, 4: <= 96, 8: <= 128, 16: above) whose kernel issues its LDS fragment reads ahead of the MFMAs
void flash_attn_set_vtr(int v);    // option "flash_vtr": same bits: row-major V tiles read with the transposing LDS read (ds_read_b64_tr_b16)
void flash_attn_set_ovl(int v);    // option "flash_ovl": 1 = overlapped issue order in the two-block d = 40 kernel (default), 2 = also the other d <= 48 two-block launches, 0 = phase by phase
void flash_attn_set_nsel(int v);   // option "flash_nsel": 1 = select-free K / V staging in the d = 40 two-block, d = 64 and d = 128 kernels (default since round 4: bit-identical, -3..6 % per launch)
void flash_attn_set_sm(int v);     // option "flash_sm": softmax arithmetic variant of the d = 64 / d = 128 one-block kernels (2 = accumulator-initialised max, 4 = v_dot2 row sums, 6 = both)
void flash_attn_set_pk(int v);     // option "flash_pk": 1 = running-max subtraction and row sums as packed f32 operations (two scores per instruction; measured slower); 0 = scalar (default)
void flash_attn_set_qb64(int v);   // option "flash_qb64": N > 0 = two query blocks per wave at d = 64 for launches with at least N workgroups of 256 queries; 0 = off
void flash_attn_set_short(int v);  // option "flash_short": k_flash_short (K / V register-resident) for 64 < Lk <= 96, d <= 64: 0 = off, 1 = on, 2 = with the next block's Q prefetched (default)
void gemm16_set_t256p_min_nt_sk(int v);     // option "t256p_min_nt_sk" (64): least 32-wide K stages of a Linear that stream-K could run for it to take the 256 x 256 tile
void gemm16_set_t256p_min_tiles_sk(int v);  // option "t256p_min_tiles_sk" (192): least tiles, same condition
bool gemm16_split_col_supported(int64_t rows, int64_t M, int64_t K);  // a Linear of this shape may carry Epilogue::split_col (it takes the pipelined 256 x 256 tile, no K slices)
void gemm16_set_streamk(int v);    // option "streamk" (0; 1 = launches of two rounds or more, 2 = every candidate): Linears whose tile count leaves the last round of a one-workgroup-per-CU tile mostly empty run as one round of persistent workgroups over equal (tile, K-tile) ranges
void gemm16_set_swp(int v);        // option "gemm16_swp": 1 = transposed-accumulator epilogue for the big-token Linear tiles (measured round 4: correct, 1 % slower per step; default 0)
void flash_attn_set_pp_min_tiles(int v);  // option "flash_pp_min_tiles"
void flash_attn_set_mslot64(int v);  // option "flash_mslot64" (0): d = 64 launches with the running max in a padded k-slot (1) and the row sums in a ones column of V (2)
void flash_attn_set_mslot(int v);  // option "flash_mslot"
void launch_flash_attn(hipStream_t s, const FlashOut& out, const View4& q, const View4& k, const View4& v, float scale);


// calib.hip: what this box delivers (bench.py roofline.measured_peaks)
struct CalibrationResult {
    float mfma_f16_tflops = 0.f, mfma_clock_mhz = 0.f, copy_tbs = 0.f, read_tbs = 0.f;
    int compute_units = 0;
};
bool calibrate_device(hipStream_t s, CalibrationResult* out);

}  // namespace mi355x
