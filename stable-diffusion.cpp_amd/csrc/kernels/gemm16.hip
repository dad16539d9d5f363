// gemm16.hip — second-generation dense contraction for gfx950: BOTH operands arrive as f16 images and are
// streamed HBM/L2 -> LDS by the LDS-DMA engine (global_load_lds_dwordx4), so the main loop contains nothing but
// DMA issue, ds_read_b128 fragment reads and v_mfma_f32_32x32x16_f16.  No f32->f16 conversion, no address math
// per element, no VGPR staging — the first-generation kernels in wgemm.hip were bound by exactly those
// (profiles/r01a_*: 3 % of the MFMA peak).
//
//   A  activations, f16 row-major [rows][Kp] (K contiguous, Kp % 64 == 0), written once by the PRODUCER of the
//      tensor (norm / GEGLU / pack kernels below) into the backend's private operand arena;
//      linear: rows = tokens;  conv: rows = output positions, the A-tile of K-tile (ic-block, tap) is gathered
//      straight from the NHWC image at (oh*S+kh-pad, ow*S+kw-pad) — implicit GEMM with K = ICp*KS*KS ordered (ic-block, tap, ic), padding
//      taps read a zero page, the nearest-x2 upsample is an index shift.  No im2col, no halo patch.
//   W  static weights in MFMA fragment order [col/32][Kp/16][64 lanes][8 halfs] (wgemm.hip: launch_wswz_*).
//
// Tile 128 rows x BN cols x 64 k per stage, 2 LDS stages (32 KB A + 32 KB W at BN=128), 4 waves (2x2, 64x64 each).
// LDS image of A: row r holds its 8 16-byte k-slots XOR-permuted by ((r>>1)&7); the permutation is applied on the
// DMA SOURCE address (the LDS destination of an LDS-DMA is always lane-linear) and again on the fragment read, which
// makes every ds_read_b128 of 16 consecutive rows hit 16 distinct 4-bank slots (cdna_hip_programming.md rule 21).
// One __syncthreads per K-tile: the barrier's implicit vmcnt(0) retires this tile's DMA, the next tile's DMA is
// issued right after it and overlaps the 32 MFMAs of the current tile; 2 workgroups per CU hide the rest.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>
#include "device_utils.h"
#include "kernels.h"
#include "ktime.h"
#include "g16_common.h"

namespace mi355x {

// geometry: workgroup tile BM x BN, WR x WC waves, each wave owns (BM/WR) x (BN/WC) outputs = RB x CB blocks of 32x32
// PIPE = 1 (tile configuration T320: 256 x 320, 8 waves of 64 x 160, BK 32, FOUR stages = 144 KB of LDS, one workgroup per CU):
// the software-pipelined main loop.  The fragment reads of the NEXT k-step are issued before the MFMAs of the current one (A fragments
// double-buffered, each B fragment re-loaded in place right behind its last use), so the matrix pipe never waits for LDS; the one
// barrier per stage sits at the head of the stage's LAST k-step — every wave has its fragments of that k-step in registers by then, so
// the barrier both publishes stage kt+1 (each wave waited for its own DMA pieces first) and frees slot kt for the DMA of stage kt+4,
// with two further stages (72 KB) in flight across it.  FLOP per DMA byte is 1.45x the 256x160 tile's (the LDS-DMA stream, ~23 B/clk/CU,
// is what bounds these kernels: profiles/r02a_gemm_ablation_kernel_stats.csv).
// SK = true (pipelined Linear tiles only): STREAM-K.  The launch is one workgroup per CU (g.sk_grid of them); the (tile, K-tile) units of the whole
// GEMM are cut into g.sk_grid equal contiguous ranges and workgroup w walks range w tile by tile.  A tile whose K range is shared by several
// workgroups is combined by its LAST-ARRIVING part: every part dumps its raw accumulators to a slab slot (write-through stores), takes a ticket on
// the tile's counter, and the last arriver sums the parts in part order (bitwise deterministic whoever arrives last) and runs the regular
// epilogue — the in-launch split-K protocol below with a per-tile part count.  Nobody waits for another workgroup.  Why: a grid of T tiles on
// 256 CUs takes ceil(T / 256) rounds; DiT Linears have T = 192 .. 1428 (FLUX 4096 x 3072 -> 9216: 576 tiles = 2.25 rounds paid as 3; SD3.5
// 8192 x 9728 -> 2432: 320 tiles = 1.25 rounds paid as 2), profiles/r05b_*.
// QT = 8 / 4 (pipelined 256 x 256 Linear tile only): the weight operand is NOT an f16 image — g.W points at the RAW GGUF q8_0 / q4_0 rows and the blocks are
// dequantised inside the main loop, once per workgroup, into the B stage the MFMA fragments are read from (DESIGN.md section 3.2 "in-loop dequantisation"):
//   * raw ring: the 2-block piece (68 / 36 B) of each of the tile's 256 weight rows that covers TWO K stages is fetched by LDS-DMA into one of three raw slots —
//     4 / 2 16-byte pieces per row [row][64 / 32 B] (piece order XOR-permuted per row on the DMA source so that the 16-byte reads below are conflict-free)
//     plus one 16-byte piece holding the LAST 16 bytes of the pair [row][16 B]: every fetched byte belongs to the row, nothing is read past a row's end;
//   * B stages: TWO f16 slots in the fragment order of the weight image; thread (column c = tid & 255, half h = tid >> 8) reads its 16 quants + the block
//     scale out of the raw slot (two ds_read_b128 + one ds_read_b32), converts them with the v_perm / v_pk_add / v_pk_mul sequence of k_qgemm16 — f16(d * q),
//     bit-identical to the value the f16 image holds — and stores the two 16-byte fragment rows with ds_write_b128;
//   * schedule (iteration kt computes stage kt): raw ds_reads for stage kt + 2 are issued in the first k-step (they return under the existing lgkmcnt(0) in
//     front of the barrier), the conversion runs in the MFMA shadow of the second k-step and its ds_writes land in the B slot stage kt just left; the raw
//     chunk of stages kt + 6 / kt + 7 is issued every even iteration behind the A pieces of stage kt + 4.  The LDS-DMA queue retires in order, so ONE counted
//     wait per iteration — vmcnt(2 A + 2 A + raw pieces) — covers both the A stage and the raw chunk that the next barrier publishes.
// The result is bitwise the same as the image path's on the same tile (same operand values, same summation order).
template <int BM, int BN, bool CONV, int BK, int NST, int WR, int WC, int PIPE = 0, bool SWP = false, bool SK = false, int QT = 0>
__global__ __launch_bounds__(WR * WC * 64, PIPE ? 2 : 2) void k_gemm16(G16Args g) {
    static_assert(!SWP || !CONV, "SWP: the Linear kernels with the accumulator transposed (g16_common.h, epi_linear_swp)");
    static_assert(!SK || (PIPE == 1 && !CONV && !SWP), "stream-K is written for the pipelined Linear tiles");
    static_assert(QT == 0 || ((QT == 8 || QT == 4) && PIPE == 1 && !CONV && !SWP && !SK && BM == 256 && (BN == 256 || BN == 192) && WR * WC == 8), "in-loop dequantisation: pipelined 256 x 256 / 256 x 192 Linear tiles");
    constexpr int NW  = WR * WC;
    constexpr int RB  = BM / WR / 32;  // 32-row blocks per wave
    constexpr int CB  = BN / WC / 32;  // 32-col blocks per wave
    static_assert(RB * WR * 32 == BM && CB * WC * 32 == BN, "tile must split into 32x32 blocks per wave");
    constexpr int ROWB   = BK * 2;     // bytes per A row in a stage (128 or 64)
    constexpr int SLOTS  = ROWB / 16;  // 16-byte k-slots per row (8 or 4)
    constexpr int RPP    = 1024 / ROWB;  // rows per 1-KiB DMA piece (8 or 16)
    constexpr int ABYTES = BM * ROWB;
    constexpr int KSTEPS = BK / 16;
    constexpr int NF     = (BN / 32) * KSTEPS;         // W fragments (1 KiB each) per stage
    constexpr int APW    = (ABYTES / 1024) / NW;       // A pieces per wave per stage
    constexpr int WPW    = (NF + NW - 1) / NW;         // W fragments per wave per stage (wave w fetches fragments w, w + NW, ...)
    constexpr int WEXTRA = NF % NW;                    // != 0: only waves < WEXTRA fetch WPW fragments, the others WPW - 1
    constexpr int BBYTES = NF * 1024;
    constexpr int NPT    = APW + WPW;                  // LDS-DMA instructions per wave per stage (waves >= WEXTRA: one fewer when WEXTRA != 0)
    static_assert(APW * NW * 1024 == ABYTES && APW >= 1, "A stage must split evenly over the waves");
    // in-loop dequantisation (QT): A ring (NST slots) | two f16 B slots | three raw slots
    constexpr int QBLK  = QT == 8 ? 34 : 18;            // bytes of a quantised block (32 weights)
    constexpr int QPB   = QT == 8 ? 64 : 32;            // bytes per row of the raw slot's piece region (whole 16-byte pieces of the 2-block pair)
    constexpr int QNP   = QPB / 16;                     // ... pieces per row
    constexpr int QTOFF = 2 * QBLK - 16;                // source offset of the tail piece (the last 16 bytes of the pair)
    constexpr int QVC   = 256;                          // columns a raw slot is laid out for: the 192-column tile fetches 64 clamped duplicates (every wave issues the same number of LDS-DMA instructions)
    constexpr int QRAWB = QVC * (QPB + 16);             // bytes of a raw slot
    constexpr int QNRAW = 3;
    constexpr int QPI   = QT ? QVC * QNP / (NW * 64) : 0;  // piece-region LDS-DMA instructions per wave and chunk (2 / 1), + 1 for the tail region
    constexpr int QRQ   = QPI + 1;
    constexpr int QB0   = NST * ABYTES;                 // first B slot
    constexpr int QR0   = QB0 + 2 * BBYTES;             // first raw slot
    constexpr int SMEMB = QT ? QR0 + QNRAW * QRAWB : NST * (ABYTES + BBYTES);
    static_assert(SMEMB <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) char smem[SMEMB];

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave % WR, wc = wave / WR;

    // stream-K: this workgroup's unit range [sk_u, sk_end) of the g.sk_tiles * g.nt (tile, K-tile) units
    int64_t sk_u = 0, sk_end = 0;
    int sk_w = 0;  // logical workgroup index: the workgroups of one XCD (blockIdx % 8) take ADJACENT unit ranges, so that — like the XCD-aware tile order of the
                   // plain launch — the 32 workgroups sharing an L2 walk tiles of the same one or two row tiles (first run without it: FLUX Linears +9 %)
    int dp_t = 0, dp_end = 0;  // hybrid: whole tiles [dp_t, dp_end) of this workgroup, walked AFTER its share of the cut tiles
    if constexpr (SK) {
        // tiles [0, sk_dp) are whole tiles, sk_dp / sk_grid per workgroup (data-parallel rounds); tiles [sk_dp, sk_tiles) — the part that does not fill a
        // round — are cut over K into equal unit ranges and come FIRST: their slab traffic then overlaps the whole-tile work of the other workgroups
        // instead of arriving as one burst at the end of the launch (sk_dp = 0: pure stream-K)
        const int64_t U = (int64_t)(g.sk_tiles - g.sk_dp) * g.nt;
        sk_w            = (g.sk_grid & 7) == 0 ? (int)(blockIdx.x & 7) * (g.sk_grid >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        sk_u            = (int64_t)sk_w * U / g.sk_grid;
        sk_end          = (int64_t)(sk_w + 1) * U / g.sk_grid;
        const int per   = g.sk_dp / g.sk_grid;
        dp_t            = sk_w * per;
        dp_end          = dp_t + per;
    }
  for (bool sk_more = true; sk_more;) {  // one pass per tile segment (stream-K); exactly one pass otherwise
    // stream-K: the lane id is made opaque per pass, so that every lane-derived address (DMA sources, fragment offsets, epilogue offsets) is
    // recomputed inside the pass it is used in instead of being hoisted to kernel entry and kept live (= spilled) across the whole loop
    int lane = lane0;
    if constexpr (SK) asm volatile("" : "+v"(lane));
    // XCD-aware tile order: consecutive ids on one XCD share the A row tile (all column tiles of a row tile)
    int bid = blockIdx.x, kslice = blockIdx.y;
    // split-K: this workgroup accumulates K tiles [kt0, kt0 + nt) and stores raw partial sums into its slab
    int kt0 = 0, nt = g.nt;
    if constexpr (SK) {
        if (sk_u < sk_end) {
            const int rt = (int)(sk_u / g.nt);  // tile inside the cut part
            bid          = g.sk_dp + rt;
            kt0          = (int)(sk_u - (int64_t)rt * g.nt);
            nt           = (int)min((int64_t)(g.nt - kt0), sk_end - sk_u);
            sk_u += nt;
        } else if (dp_t < dp_end) {
            bid = dp_t++;
        } else {
            break;  // more workgroups than units
        }
        sk_more = sk_u < sk_end || dp_t < dp_end;
    } else {
        sk_more = false;
        g16_wg_order(g, g.multi > 1 ? g.ncol_tiles * g.multi : g.ncol_tiles, bid, kslice);
#ifdef MI355X_EXPERIMENTS
        if (g.abl == 7) nt = PIPE ? 4 : 1;  // timing ablation: (almost) no main loop, the launch's fixed cost + epilogue
#endif
        if (g.split_k > 1) {
            kt0 = kslice * g.nt_slice;
            nt  = min(g.nt_slice, g.nt - kt0);
            if (!g.sk_cnt) g.dst += (int64_t)kslice * g.slab;
        }
    }
    const int nct_all  = g.multi > 1 ? g.ncol_tiles * g.multi : g.ncol_tiles;
    const int row_tile = bid / nct_all;
    int col_tile       = bid - row_tile * nct_all;
    if (!CONV && g.multi > 1) {  // workgroup-uniform: pick this tile's weight, destination and bias
        const int wi = col_tile / g.ncol_tiles;
        col_tile -= wi * g.ncol_tiles;
        // static indices + scalar selects: indexing the kernel-argument struct with a run-time value makes the compiler copy ALL of it to
        // scratch memory (392 B per lane, +50 VGPRs in every Linear instantiation: r02y, Linear family 7.5 -> 11.6 ms)
#define G16_PICK(F, A)                                                                                                                                    \
    g.F = wi < 8 ? (wi < 4 ? (wi < 2 ? (wi == 0 ? g.A[0] : g.A[1]) : (wi == 2 ? g.A[2] : g.A[3])) : (wi < 6 ? (wi == 4 ? g.A[4] : g.A[5]) : (wi == 6 ? g.A[6] : g.A[7]))) \
                 : (wi < 12 ? (wi < 10 ? (wi == 8 ? g.A[8] : g.A[9]) : (wi == 10 ? g.A[10] : g.A[11]))                                                       \
                            : (wi < 14 ? (wi == 12 ? g.A[12] : g.A[13]) : (wi == 14 ? g.A[14] : g.A[15])))
        G16_PICK(W, Wm);
        G16_PICK(dst, dstm);
        G16_PICK(dst16, dst16m);
        G16_PICK(ep.bias, biasm);
#undef G16_PICK
    }
    const int64_t row0 = (CONV ? 0 : g.row_base) + (int64_t)row_tile * BM;
    const int col0     = col_tile * BN;

    // XOR permutation of the k-slots of LDS row r (conflict-free ds_read_b128 over 16 consecutive rows)
    auto rowswz = [](int r) { return SLOTS == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

    // ---- per-lane DMA sources.  A: wave w issues pieces i = w*APW+q: rows RPP*i + lane/SLOTS, physical slot lane%SLOTS
    const _Float16* asrc[APW];   // rows mode: row pointer (+slot); conv: pointer of the tap-(0,0) pixel (+slot), may be out of range
    unsigned a_mask[APW];        // conv: bit t set <=> tap t of this output position reads a real pixel (else the zero page)
    int a_slot[APW];
#pragma unroll
    for (int q = 0; q < APW; ++q) {
        const int r  = (wave * APW + q) * RPP + lane / SLOTS;
        const int ls = (lane % SLOTS) ^ rowswz(r);  // logical k-slot fetched into physical slot lane%SLOTS
        a_slot[q]    = ls * 8;
        int64_t row  = row0 + r;
        if (!CONV) {
            if (row >= g.R) row = g.R - 1;
            if (g.a_runL > 0) row = (int64_t)((uint32_t)row / (uint32_t)g.a_runL) * g.a_runS + (uint32_t)row % (uint32_t)g.a_runL;  // rows in runs (token slices of a wider image)
            asrc[q] = g.A + row * g.lda + ls * 8;
        } else {
            // all per-position address work happens ONCE here; the K loop only adds a wave-uniform tap offset
            const bool ok = row < g.R;
            if (!ok) row = g.R - 1;
            const int img = (int)(row / g.OHOW);
            const int p   = (int)(row - (int64_t)img * g.OHOW);
            const int oh = p / g.OW, ow = p - oh * g.OW;
            const int CH = g.UPS ? g.H * 2 : g.H, CW = g.UPS ? g.Wd * 2 : g.Wd;
            unsigned m = 0;
            for (int t = 0; t < g.KS * g.KS; ++t) {
                const int ih = oh * g.S + t / g.KS - g.pad, iw = ow * g.S + t % g.KS - g.pad;
                if (ok && ih >= 0 && ih < CH && iw >= 0 && iw < CW) m |= 1u << t;
            }
            a_mask[q] = m;
            if (!g.UPS) {
                // tap (kh,kw) pixel = base + ((kh*Wd + kw) * ICp) halfs  (no upsample: source index is affine in the tap)
                asrc[q] = g.A + (((int64_t)img * g.H + (oh * g.S - g.pad)) * g.Wd + (ow * g.S - g.pad)) * g.ICp + ls * 8;
            } else {
                // nearest x2: keep (img, oh, ow) packed; resolved per tap (3 taps map to 2 source pixels)
                asrc[q] = g.A + ((int64_t)img * g.H * g.Wd) * g.ICp + ls * 8;
                a_slot[q] |= (oh << 8) | (ow << 20);  // oh, ow < 4096
            }
        }
    }
    const half8_t* wsrc[WPW];
    int wdst[WPW];
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
        const int f  = q * NW + wave;
        const int fs = f < NF ? f : NF - 1;
        const int cb = fs / KSTEPS, ks = fs % KSTEPS;
        int wb       = col0 / 32 + cb;
        if (!CONV && g.wblk_lim > 0 && wb >= g.wblk_lim) wb = g.wblk_lim - 1;  // column blocks past the padded image: any valid block (their outputs are masked by col < C)
        wsrc[q]      = QT ? g.W : g.W + ((int64_t)wb * g.kfr + ks) * 64 + lane;  // (QT: g.W points at raw quantised rows, fetched by the loop further down)
        wdst[q]      = fs * 1024;
    }
    const bool w_short = WEXTRA != 0 && wave >= WEXTRA;  // this wave issues one W fragment fewer per stage
    const int ktiles_per_icb = 64 / BK;  // conv: K tiles per 64-channel block (1 or 2)

    // conv tile cursor: K tile -> (64-channel block icb, tap = (kh, kw), half sub).  Tiles are staged strictly in order, so the cursor
    // advances by carries — the per-tile integer divisions this replaces were ~60 scalar instructions in front of every DMA issue.
    int c_sub = 0, c_tap = 0, c_icb = 0, c_kh = 0, c_kw = 0;
    if (CONV) {
        const int kb = kt0 / ktiles_per_icb, ntaps = g.KS * g.KS;
        c_sub        = kt0 - kb * ktiles_per_icb;
        if (g.tap_major) {
            c_tap = kb / g.icb_per_tap;
            c_icb = kb - c_tap * g.icb_per_tap;
        } else {
            c_icb = kb / ntaps;
            c_tap = kb - c_icb * ntaps;
        }
        c_kh = c_tap / g.KS;
        c_kw = c_tap - c_kh * g.KS;
    }
// advance the cursor by one K tile (plain statements on locals, NOT a lambda: a mutating by-reference capture made the compiler keep the
// cursor in scratch memory and in VGPRs)
#define G16_ADVANCE()                                  \
    do {                                               \
        if (CONV && ++c_sub >= ktiles_per_icb) {       \
            c_sub = 0;                                 \
            if (g.tap_major) {                         \
                if (++c_icb >= g.icb_per_tap) {        \
                    c_icb = 0;                         \
                    ++c_tap;                           \
                    if (++c_kw == g.KS) {              \
                        c_kw = 0;                      \
                        ++c_kh;                        \
                    }                                  \
                }                                      \
            } else {                                   \
                ++c_tap;                               \
                if (++c_kw == g.KS) {                  \
                    c_kw = 0;                          \
                    ++c_kh;                            \
                }                                      \
                if (c_tap == g.KS * g.KS) {            \
                    c_tap = c_kh = c_kw = 0;           \
                    ++c_icb;                           \
                }                                      \
            }                                          \
        }                                              \
    } while (0)
    auto stage = [&](int kt, int buf, int tap, int kh, int kw, int icb, int sub) {
        char* sa = smem + buf * (ABYTES + BBYTES);
        char* sb = sa + ABYTES;
        if (!CONV) {
#pragma unroll
            for (int q = 0; q < APW; ++q) GLDS16(asrc[q] + (int64_t)kt * BK, sa + (wave * APW + q) * 1024);
        } else {
            // K order = (64-channel block, tap, channel): the KS*KS taps of one channel block are consecutive K tiles, so a workgroup
            // re-reads the same small input window (rows +-1, 128 B per pixel) back to back and the re-reads hit L2 (tap-major
            // order re-read the whole tile 9 times, one full K sweep apart: 3x the algorithmic HBM/MALL fetch, profiles/r01c)
            // (option "conv_tap_major" keeps the old (tap, channel block) order for A/B runs)
            const int64_t koff = (int64_t)icb * 64 + sub * BK;  // wave-uniform (SALU)
            if (!g.UPS) {
                const int64_t toff = ((int64_t)kh * g.Wd + kw) * g.ICp + koff;
#pragma unroll
                for (int q = 0; q < APW; ++q) {
                    const _Float16* p = (((a_mask[q] >> tap) & 1u) && !(PIPE == 4 && tap != 0)) ? asrc[q] + toff : g.zero + (a_slot[q] & 63);
                    GLDS16(p, sa + (wave * APW + q) * 1024);
                }
            } else {
#pragma unroll
                for (int q = 0; q < APW; ++q) {
                    const int oh = (a_slot[q] >> 8) & 4095, ow = (a_slot[q] >> 20) & 4095;
                    const int sy = (oh * g.S + kh - g.pad) >> 1, sx = (ow * g.S + kw - g.pad) >> 1;
                    const _Float16* p = ((a_mask[q] >> tap) & 1u) ? asrc[q] + ((int64_t)sy * g.Wd + sx) * g.ICp + koff : g.zero + (a_slot[q] & 63);
                    GLDS16(p, sa + (wave * APW + q) * 1024);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < WPW; ++q)
            if (q + 1 < WPW || !w_short) GLDS16(wsrc[q] + (int64_t)kt * KSTEPS * 64, sb + wdst[q]);
    };

    // the same, ONE LDS-DMA piece at a time (idx < APW: A piece idx, else W fragment idx - APW; same issue order as stage()) — the pipelined loop
    // interleaves the pieces with its MFMA groups so that the address arithmetic issues in the shadow of the wave's own MFMAs
    auto stage_piece = [&](int idx, int kt, int buf, int tap, int kh, int kw, int icb, int sub) {
        char* sa = smem + buf * (ABYTES + BBYTES);
        char* sb = sa + ABYTES;
        if (idx < APW) {
            const int q = idx;
            if (!CONV) {
                GLDS16(asrc[q] + (int64_t)kt * BK, sa + (wave * APW + q) * 1024);
            } else {
                const int64_t koff = (int64_t)icb * 64 + sub * BK;
                if (!g.UPS) {
                    const int64_t toff = ((int64_t)kh * g.Wd + kw) * g.ICp + koff;
                    const _Float16* p  = (((a_mask[q] >> tap) & 1u) && !(PIPE == 4 && tap != 0)) ? asrc[q] + toff : g.zero + (a_slot[q] & 63);  // PIPE 4: ablation
                    GLDS16(p, sa + (wave * APW + q) * 1024);
                } else {
                    const int oh = (a_slot[q] >> 8) & 4095, ow = (a_slot[q] >> 20) & 4095;
                    const int sy = (oh * g.S + kh - g.pad) >> 1, sx = (ow * g.S + kw - g.pad) >> 1;
                    const _Float16* p = ((a_mask[q] >> tap) & 1u) ? asrc[q] + ((int64_t)sy * g.Wd + sx) * g.ICp + koff : g.zero + (a_slot[q] & 63);
                    GLDS16(p, sa + (wave * APW + q) * 1024);
                }
            }
        } else {
            const int q = idx - APW;
            if (q < WPW && (q + 1 < WPW || !w_short)) GLDS16(wsrc[q] + (int64_t)kt * KSTEPS * 64, sb + wdst[q]);
        }
    };

    float16_t acc[RB][CB];
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) acc[a][b] = (float16_t){0};

    // fragment read offsets (bytes) inside a stage
    int aoff[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) aoff[rb] = (wr * (RB * 32) + rb * 32 + (lane & 31)) * ROWB;
    const int aswz = rowswz(lane & 31);  // row blocks start at multiples of 32 rows -> the permutation depends on the lane only
    const int hi   = lane >> 5;

    auto compute = [&](int buf) {
        const char* sa = smem + buf * (ABYTES + BBYTES);
        const char* sb = sa + ABYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            half8_t af[RB], bf[CB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) af[rb] = *(const half8_t*)(sa + aoff[rb] + (((ks * 2 + hi) ^ aswz) << 4));
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) bf[cb] = *(const half8_t*)(sb + (((wc * CB + cb) * KSTEPS + ks) * 64 + lane) * 16);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    if (CONV || SWP)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cb], af[rb], acc[rb][cb], 0, 0, 0);  // D[oc][pos]
                    else
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rb], bf[cb], acc[rb][cb], 0, 0, 0);  // D[row][col]
                }
        }
    };

    if constexpr (PIPE) {
        // ---- hand-scheduled software pipeline (see the header comment of this kernel).  Fragment reads and waits are inline asm: left to
        // the compiler, every k-step opened with s_waitcnt lgkmcnt(0) behind freshly issued reads and the reads were sunk below the MFMAs.
        // Ordering rules used here: asm volatile statements keep their program order; an MFMA is tied behind a wait by passing its operand
        // registers through an empty asm ("+v") placed after the wait; sched_barrier(0) keeps the {MFMA group, read} interleave as written.
        static_assert(KSTEPS == 2 && NST == 4 && RB == 2 && ROWB == 64 && CB >= 3, "the pipelined loop is written for BK 32 x 4 stages, 64-row waves");
        constexpr int STAGE = ABYTES + BBYTES;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        const uint32_t aad0 = lds0 + (uint32_t)aoff[0] + (uint32_t)(((0 + hi) ^ aswz) << 4);  // row block 0, k-step 0 (row block 1: + 32 rows = + 2048 B)
        const uint32_t aad1 = lds0 + (uint32_t)aoff[0] + (uint32_t)(((2 + hi) ^ aswz) << 4);  // k-step 1
        const uint32_t bad  = lds0 + (uint32_t)(ABYTES + (wc * CB * KSTEPS) * 1024 + lane * 16);
        // fragment registers: two A sets (current / next k-step); column blocks 0 .. CB-3 are re-read in place right behind their MFMAs;
        // the LAST two column blocks are double-buffered (BH0 / BH1) and read at the head of a k-step, so that no read is issued during
        // the last two MFMA groups (128 cycles) in front of the wait that needs every fragment of the next k-step
        constexpr int CL = CB - 2;
        half8_t A0[RB], A1[RB], BL[CL], BH0[2], BH1[2];
        // PIPE == 2 / 3 exist only in -DMI355X_EXPERIMENTS builds (scripts/gemm_ablation.py): wrong-result TIMING ablations — 2 keeps the
        // DMA stream, reads, waits and barriers but issues no MFMA; 3 keeps reads + MFMAs but stages nothing after the pipeline fill
        auto mma = [&](int rb, int cb, const half8_t& a, const half8_t& b) {
            if constexpr (PIPE == 2) {
                asm volatile("" ::"v"(a), "v"(b));
                return;
            }
            if (CONV || SWP)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[rb][cb], 0, 0, 0);  // D[oc][pos]
            else
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[rb][cb], 0, 0, 0);  // D[row][col]
        };
#define G16_RD(DST_, ADDR_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_))
#define G16_TIE(X_) asm volatile("" : "+v"(X_))
        // One k-step.  ACUR / ANXT, HCUR / HNXT: current and next fragment sets; an_ / bn_: LDS addresses of this lane's A row (row block 0)
        // and of this wave's first B fragment for the NEXT k-step (KSN_ = its index inside its stage)
#define G16_NOHOOK(I_) ((void)0)
#define G16_KSTEP(ACUR, ANXT, HCUR, HNXT, an_, bn_, KSN_) G16_KSTEP_H(ACUR, ANXT, HCUR, HNXT, an_, bn_, KSN_, G16_NOHOOK)
        // HOOK_(i), i = 0 .. 4: a statement issued behind MFMA group i (the steady-state loop issues one LDS-DMA piece of stage kt+4 there)
#define G16_KSTEP_H(ACUR, ANXT, HCUR, HNXT, an_, bn_, KSN_, HOOK_)                                                   \
    do {                                                                                                             \
        G16_RD(ANXT[0], an_, 0);                                                                                     \
        G16_RD(ANXT[1], an_, 2048);                                                                                  \
        G16_RD(HNXT[0], bn_, ((CB - 2) * KSTEPS + (KSN_)) * 1024);                                                   \
        G16_RD(HNXT[1], bn_, ((CB - 1) * KSTEPS + (KSN_)) * 1024);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        mma(0, 0, ACUR[0], BL[0]);                                                                                   \
        mma(1, 0, ACUR[1], BL[0]);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        G16_RD(BL[0], bn_, (0 * KSTEPS + (KSN_)) * 1024);                                                            \
        HOOK_(0);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if constexpr (CL > 1) {                                                                                      \
            mma(0, 1, ACUR[0], BL[CL > 1 ? 1 : 0]);                                                                  \
            mma(1, 1, ACUR[1], BL[CL > 1 ? 1 : 0]);                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            G16_RD(BL[CL > 1 ? 1 : 0], bn_, (1 * KSTEPS + (KSN_)) * 1024);                                           \
        }                                                                                                            \
        HOOK_(1);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if constexpr (CL > 2) {                                                                                      \
            mma(0, 2, ACUR[0], BL[CL - 1]);                                                                          \
            mma(1, 2, ACUR[1], BL[CL - 1]);                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            G16_RD(BL[CL - 1], bn_, (2 * KSTEPS + (KSN_)) * 1024);                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        HOOK_(2);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        mma(0, CB - 2, ACUR[0], HCUR[0]);                                                                            \
        mma(1, CB - 2, ACUR[1], HCUR[0]);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        HOOK_(3);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        mma(0, CB - 1, ACUR[0], HCUR[1]);                                                                            \
        mma(1, CB - 1, ACUR[1], HCUR[1]);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        HOOK_(4);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
        // tie every fragment the next MFMAs use behind the s_waitcnt issued just before (asm volatile statements keep their order)
#define G16_TIE_FRAGS(AS_, HS_)                                                                                      \
    do {                                                                                                             \
        G16_TIE(AS_[0]);                                                                                             \
        G16_TIE(AS_[1]);                                                                                             \
        _Pragma("unroll") for (int cb = 0; cb < CL; ++cb) G16_TIE(BL[cb]);                                           \
        G16_TIE(HS_[0]);                                                                                             \
        G16_TIE(HS_[1]);                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#define G16_DMA_HOOK(I_)                                                                                             \
    do {                                                                                                             \
        if (PIPE != 3 && PIPE != 5 && PIPE != 6 && (I_) < NPT) stage_piece((I_), kt0 + kt + NST, fbuf, c_tap, c_kh, c_kw, c_icb, c_sub);      \
        if (PIPE != 3 && PIPE != 5 && PIPE != 6 && (I_) == 4 && NPT > 5) {                                                                     \
            _Pragma("unroll") for (int e_ = 5; e_ < NPT; ++e_) stage_piece(e_, kt0 + kt + NST, fbuf, c_tap, c_kh, c_kw, c_icb, c_sub); \
        }                                                                                                            \
    } while (0)
#define G16_PIPE_LOOP(NP_)                                                                                                         \
    do {                                                                                                                           \
        const int npro = nt < NST ? nt : NST;                                                                                      \
        for (int i = 0; i < npro; ++i) {                                                                                           \
            stage(kt0 + i, i, c_tap, c_kh, c_kw, c_icb, c_sub);                                                                    \
            G16_ADVANCE();                                                                                                         \
        }                                                                                                                          \
        /* stage 0 landed (later stages stay in flight), published by the barrier */                                              \
        if (npro >= 4)                                                                                                             \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NP_)) : "memory");                                                       \
        else if (npro == 3)                                                                                                        \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NP_)) : "memory");                                                       \
        else if (npro == 2)                                                                                                        \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP_) : "memory");                                                             \
        else                                                                                                                       \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                       \
        asm volatile("s_barrier" ::: "memory");                                                                                    \
        G16_RD(A0[0], aad0, 0);                                                                                                    \
        G16_RD(A0[1], aad0, 2048);                                                                                                 \
        G16_RD(BL[0], bad, (0 * KSTEPS) * 1024);                                                                                   \
        if constexpr (CL > 1) G16_RD(BL[CL > 1 ? 1 : 0], bad, (1 * KSTEPS) * 1024);                                                \
        if constexpr (CL > 2) G16_RD(BL[CL - 1], bad, (2 * KSTEPS) * 1024);                                                        \
        G16_RD(BH0[0], bad, ((CB - 2) * KSTEPS) * 1024);                                                                           \
        G16_RD(BH0[1], bad, ((CB - 1) * KSTEPS) * 1024);                                                                           \
        int buf = 0, kt = 0;                                                                                                       \
        /* steady state: stages kt+1 .. kt+3 exist and stage kt+4 is issued — nothing in the body is conditional */               \
        for (; kt + NST < nt; ++kt) {                                                                                              \
            const uint32_t so = (uint32_t)buf * STAGE;                                                                             \
            if constexpr (PIPE != 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
            G16_TIE_FRAGS(A0, BH0);                                                                                                \
            G16_KSTEP(A0, A1, BH0, BH1, aad1 + so, bad + so, 1);                                                                   \
            /* this wave's pieces of stage kt+1 have landed (two younger stages stay in flight) and all its fragment reads of       \
               stage kt are complete; the barrier then makes stage kt+1 readable and this slot refillable for everyone */           \
            if constexpr (PIPE != 6) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NP_)) : "memory");                   \
            G16_TIE_FRAGS(A1, BH1);                                                                                                \
            if constexpr (PIPE != 5 && PIPE != 6) asm volatile("s_barrier" ::: "memory");                                          \
            /* the slot just freed takes stage kt+4: its NPT LDS-DMA pieces are issued one per MFMA group of this k-step */         \
            const int fbuf    = buf;                                                                                               \
            buf               = buf == NST - 1 ? 0 : buf + 1;                                                                      \
            const uint32_t sn = (uint32_t)buf * STAGE;                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
            G16_KSTEP_H(A1, A0, BH1, BH0, aad0 + sn, bad + sn, 0, G16_DMA_HOOK);                                                   \
            G16_ADVANCE();                                                                                                         \
        }                                                                                                                          \
        /* drain: the last (up to NST) stages, nothing left to issue */                                                           \
        for (; kt < nt; ++kt) {                                                                                                    \
            const uint32_t so = (uint32_t)buf * STAGE;                                                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
            G16_TIE_FRAGS(A0, BH0);                                                                                                \
            G16_KSTEP(A0, A1, BH0, BH1, aad1 + so, bad + so, 1);                                                                   \
            if (kt + 3 < nt)                                                                                                       \
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NP_)) : "memory");                                        \
            else if (kt + 2 < nt)                                                                                                  \
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP_) : "memory");                                              \
            else                                                                                                                   \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                        \
            G16_TIE_FRAGS(A1, BH1);                                                                                                \
            asm volatile("s_barrier" ::: "memory");                                                                                \
            buf               = buf == NST - 1 ? 0 : buf + 1;                                                                      \
            const uint32_t sn = (uint32_t)buf * STAGE;                                                                             \
            /* after the last stage the "next" fragment reads fetch stale bytes of the following slot: harmless, never used —      \
               keeping the body unconditional keeps the accumulators in place (a branch here made the compiler copy and spill them) */ \
            G16_KSTEP(A1, A0, BH1, BH0, aad0 + sn, bad + sn, 0);                                                                   \
        }                                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                         \
    } while (0)
        if constexpr (QT != 0) {
            // ================= in-loop dequantisation (see the header comment of this kernel) =================
            typedef uint32_t qu32x4_t __attribute__((ext_vector_type(4)));
            typedef _Float16 qhalf2_t __attribute__((ext_vector_type(2)));
            const int tid = (int)threadIdx.x;
            constexpr int QWH = BN / 64;               // waves per half of the conversion pass (4; 3 on the 192-column tile, whose last two waves convert nothing)
            const int qh      = wave / QWH;            // conversion pass: this thread's half of the block's 32 weights (wave-uniform) ...
            const int qc      = (tid - qh * BN) & 255; // ... and its weight row (column of the tile)
            const bool qact   = wave < 2 * QWH;
            auto qf = [](int c) { return QT == 8 ? ((c >> 2) & 3) : ((c >> 3) & 1); };  // physical piece slot of logical piece p of raw row c: p ^ qf(c)
            const char* qsrc[QPI];
#pragma unroll
            for (int r = 0; r < QPI; ++r) {
                const int idx = r * (NW * 64) + tid, col = idx / QNP, pp = idx % QNP;
                const int64_t wr_ = min((int64_t)col0 + (col < BN ? col : BN - 1), g.C - 1);  // (a ragged last column tile re-fetches the last weight row: those outputs are masked)
                qsrc[r]       = (const char*)g.W + wr_ * g.qrow_bytes + ((pp ^ qf(col)) << 4);
            }
            const int qtc     = wave * 32 + (lane & 31);
            const char* qsrcT = (const char*)g.W + min((int64_t)col0 + (qtc < BN ? qtc : BN - 1), g.C - 1) * g.qrow_bytes + QTOFF;
            // byte offsets of this thread's raw reads inside a raw slot, per block of the pair ([0] / [1]): ONE 16-byte piece (A), the dword that follows the
            // piece's bytes in the row (B: the fifth dword of a 2-byte-misaligned run) and the dword holding the block scale (D).  A thread needs 16 quant bytes +
            // the scale, 18 bytes: it reads 20 (24 for the second half of a q8_0 pair's first block, whose scale sits 18 bytes ahead of its quants)
            uint32_t qoA[2], qoB[2], qoD[2];
            {
                const int f = qf(qc);
                auto pc = [&](int p) { return (uint32_t)(qc * QPB + ((p ^ f) << 4)); };
                const uint32_t T = (uint32_t)(QVC * QPB + qc * 16);
                if constexpr (QT == 8) {
                    // block 0: scale at bytes 0-1, quants 2..33; block 1: scale at 34-35, quants 36..67 (the tail piece holds bytes 52..67)
                    qoA[0] = qh ? pc(1) : pc(0);   // bytes 16..31 | 0..15
                    qoB[0] = qh ? pc(2) : pc(1);   // dword at 32 | 16
                    qoD[0] = pc(0);                // scale: low half of the dword at 0
                    qoA[1] = qh ? T : pc(2);       // bytes 52..67 | 32..47
                    qoB[1] = pc(3);                // (h = 0) dword at 48
                    qoD[1] = pc(2);                // scale: high half of the dword at 32
                } else {
                    // block 0: scale at 0-1, nibble bytes 2..17; block 1: scale at 18-19, nibble bytes 20..35 (= the tail piece)
                    qoA[0] = pc(0);                // bytes 0..15
                    qoB[0] = pc(1);                // dword at 16
                    qoD[0] = pc(0);
                    qoA[1] = T;                    // bytes 20..35
                    qoB[1] = pc(1);
                    qoD[1] = pc(1);                // scale: high half of the dword at 16
                }
            }
            const uint32_t qraw0 = lds0 + (uint32_t)QR0;
            const uint32_t qwB   = lds0 + (uint32_t)(QB0 + ((qc >> 5) * KSTEPS + qh) * 1024 + (qc & 31) * 16);  // first fragment row this thread writes (second: + 512)
            const uint32_t badq  = lds0 + (uint32_t)(QB0 + (wc * CB * KSTEPS) * 1024 + lane * 16);
            const int qnch       = nt >> 1;          // 2-stage chunks of this K range (kt0 and nt are even: launcher precondition)
            const int64_t qch0   = (int64_t)(kt0 >> 1) * (2 * QBLK);
            qu32x4_t RA;
            uint32_t RB = 0, RD = 0;
            auto q_issue_raw_piece = [&](int i, int ch, int slot) {  // LDS-DMA piece i (< QPI: piece region, QPI: tail region) of chunk ch into raw slot `slot`
                const int cc      = ch < qnch ? ch : qnch - 1;  // past the end: a harmless re-fetch of the last chunk (keeps the per-iteration DMA count constant)
                const int64_t off = qch0 + (int64_t)cc * (2 * QBLK);
                char* rb          = smem + QR0 + slot * QRAWB;
                if (i < QPI) {
                    GLDS16(qsrc[i < QPI ? i : 0] + off, rb + (i * (NW * 64) + wave * 64) * 16);
                } else {
                    if (lane < 32) GLDS16(qsrcT + off, rb + QVC * QPB + wave * 512);
                }
            };
            auto q_issue_a = [&](int q, int ktabs, int slot) { GLDS16(asrc[q] + (int64_t)ktabs * BK, smem + slot * ABYTES + (wave * APW + q) * 1024); };
            // registers -> the 16 quant bytes (Qv) and the block scale of block B_ of the pair
            // part 1: the 16 quant bytes of block b out of the raw registers (realigned when the run starts 2 bytes into a dword) + the scale's 16 bits
            auto q_extract1 = [&](int b, uint32_t (&Qv)[4], uint32_t& dbits) {
                if (b == 0) {  // the run starts 2 bytes into RA (scale or a neighbour's quants in front of it)
                    Qv[0] = __builtin_amdgcn_alignbit(RA[1], RA[0], 16);
                    Qv[1] = __builtin_amdgcn_alignbit(RA[2], RA[1], 16);
                    Qv[2] = __builtin_amdgcn_alignbit(RA[3], RA[2], 16);
                    Qv[3] = __builtin_amdgcn_alignbit(RB, RA[3], 16);
                    dbits = ((QT == 8 && qh) ? RD : RA[0]) & 0xffffu;
                } else if (QT == 8 && !qh) {
                    Qv[0] = RA[1];
                    Qv[1] = RA[2];
                    Qv[2] = RA[3];
                    Qv[3] = RB;
                    dbits = RA[0] >> 16;
                } else {
                    Qv[0] = RA[0];
                    Qv[1] = RA[1];
                    Qv[2] = RA[2];
                    Qv[3] = RA[3];
                    dbits = RD >> 16;
                }
            };
            // part 2: nibble select (q4_0: this thread's half of the block is the low or the high nibbles) / sign flip (q8_0: u = q + 128), scale as half2
            auto q_extract2 = [&](uint32_t (&Qv)[4], uint32_t dbits, qhalf2_t& d2) {
                if constexpr (QT == 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Qv[i] = (qh ? Qv[i] >> 4 : Qv[i]) & 0x0F0F0F0Fu;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Qv[i] ^= 0x80808080u;
                }
                const _Float16 d = __builtin_bit_cast(_Float16, (uint16_t)dbits);
                d2               = (qhalf2_t){d, d};
            };
            auto q_extract = [&](int b, uint32_t (&Qv)[4], qhalf2_t& d2) {
                uint32_t dbits;
                q_extract1(b, Qv, dbits);
                q_extract2(Qv, dbits, d2);
            };
            // four bytes -> two half2 {1024 + b0, 1024 + b1}, {1024 + b2, 1024 + b3} (exponent byte 0x64), minus 1152 / 1032 = the exact integer, times d: f16(d * q)
            auto q_deq4 = [&](uint32_t u, qhalf2_t d2, uint32_t& o01, uint32_t& o23) {
                const qhalf2_t off = {(_Float16)(QT == 8 ? 1152.f : 1032.f), (_Float16)(QT == 8 ? 1152.f : 1032.f)};
                const uint32_t p01 = __builtin_amdgcn_perm(0x64646464u, u, 0x04010400u);
                const uint32_t p23 = __builtin_amdgcn_perm(0x64646464u, u, 0x04030402u);
                o01                = __builtin_bit_cast(uint32_t, (__builtin_bit_cast(qhalf2_t, p01) - off) * d2);
                o23                = __builtin_bit_cast(uint32_t, (__builtin_bit_cast(qhalf2_t, p23) - off) * d2);
            };
    // (which of the three reads a (type, block, half) combination needs is a compile-time / wave-uniform fact: see the offset table above)
#define G16Q_RAW_READ(B_, RBASE_)                                                                                      \
    do {                                                                                                             \
        asm volatile("ds_read_b128 %0, %1" : "=v"(RA) : "v"((RBASE_) + qoA[B_]));                                    \
        if (((B_) == 0) || (QT == 8 && !qh)) asm volatile("ds_read_b32 %0, %1" : "=v"(RB) : "v"((RBASE_) + qoB[B_])); \
        if ((QT == 4 && (B_) == 1) || (QT == 8 && qh)) asm volatile("ds_read_b32 %0, %1" : "=v"(RD) : "v"((RBASE_) + qoD[B_])); \
    } while (0)
#define G16Q_WRITE(SLOT_, HALF_)                                                                                      \
    do {                                                                                                             \
        const qu32x4_t w_ = {O[4 * (HALF_)], O[4 * (HALF_) + 1], O[4 * (HALF_) + 2], O[4 * (HALF_) + 3]};           \
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(qwB + (uint32_t)((SLOT_) * BBYTES)), "v"(w_), "n"(512 * (HALF_)) : "memory"); \
    } while (0)
            // ---- fill: raw chunks 0 .. 2 and A stages 0 .. 3, in the order the steady state would have issued them
            q_issue_raw_piece(0, 0, 0);
            if constexpr (QPI > 1) q_issue_raw_piece(1, 0, 0);
            q_issue_raw_piece(QPI, 0, 0);
#pragma unroll
            for (int q = 0; q < APW; ++q) q_issue_a(q, kt0 + 0, 0);
            q_issue_raw_piece(0, 1, 1);
            if constexpr (QPI > 1) q_issue_raw_piece(1, 1, 1);
            q_issue_raw_piece(QPI, 1, 1);
#pragma unroll
            for (int q = 0; q < APW; ++q) q_issue_a(q, kt0 + 1, 1);
#pragma unroll
            for (int q = 0; q < APW; ++q) q_issue_a(q, kt0 + 2, 2);
            q_issue_raw_piece(0, 2, 2);
            if constexpr (QPI > 1) q_issue_raw_piece(1, 2, 2);
            q_issue_raw_piece(QPI, 2, 2);
#pragma unroll
            for (int q = 0; q < APW; ++q) q_issue_a(q, kt0 + 3, 3);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QRQ + 4 * APW) : "memory");  // raw chunk 0 landed
            asm volatile("s_barrier" ::: "memory");
            // conversion state carried from the second k-step of an iteration (which converts the first 8 weights of stage kt + 2) to the first k-step of the
            // next one (which converts the other 8): the last two quant dwords and the scale
            uint32_t Qv[4], qdbits = 0;
            qhalf2_t qd2;
            if (qact) {   // B stage 0 (block 0 of chunk 0) and the first half of B stage 1, outside the pipeline
                uint32_t O[8];
                G16Q_RAW_READ(0, qraw0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G16_TIE(RA);
                G16_TIE(RB);
                G16_TIE(RD);
                q_extract(0, Qv, qd2);
#pragma unroll
                for (int i = 0; i < 4; ++i) q_deq4(Qv[i], qd2, O[2 * i], O[2 * i + 1]);
                G16Q_WRITE(0, 0);
                G16Q_WRITE(0, 1);
                G16Q_RAW_READ(1, qraw0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G16_TIE(RA);
                G16_TIE(RB);
                G16_TIE(RD);
                q_extract(1, Qv, qd2);
                q_deq4(Qv[0], qd2, O[0], O[1]);
                q_deq4(Qv[1], qd2, O[2], O[3]);
                G16Q_WRITE(1, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QRQ + 3 * APW) : "memory");  // A stage 0 and raw chunk 1 landed; this thread's B writes done
            asm volatile("s_barrier" ::: "memory");
            G16_RD(A0[0], aad0, 0);
            G16_RD(A0[1], aad0, 2048);
            G16_RD(BL[0], badq, (0 * KSTEPS) * 1024);
            if constexpr (CL > 1) G16_RD(BL[CL > 1 ? 1 : 0], badq, (1 * KSTEPS) * 1024);
            if constexpr (CL > 2) G16_RD(BL[CL - 1], badq, (2 * KSTEPS) * 1024);
            G16_RD(BH0[0], badq, ((CB - 2) * KSTEPS) * 1024);
            G16_RD(BH0[1], badq, ((CB - 1) * KSTEPS) * 1024);
            int buf = 0, kt = 0, rs = 0;  // rs: raw slot of chunk kt / 2
            // first k-step of iteration kt: the raw bytes of stage kt + 2 (block qb_ of chunk kt / 2 + 1) into registers; the second 8 weights of stage kt + 1
            // (started by the previous iteration) converted and stored into B slot 1 - qb_ — the barrier of this iteration publishes that stage
#define G16Q_PIN2(A_, B_)                                                                                             \
    do {                                                                                                             \
        G16_TIE(A_);                                                                                                 \
        G16_TIE(B_);                                                                                                 \
    } while (0)
    // (every slice of the conversion is pinned to its hook by passing its results through an empty asm volatile: left alone, the compiler sinks the
    // side-effect-free VALU work down to its first use and the whole conversion runs in one lump in front of the ds_write)
#define G16Q_RD_HOOK(I_)                                                                                              \
    do {                                                                                                             \
        if ((I_) == 0 && qconv_ && qact) G16Q_RAW_READ(qb_, qraw0 + (uint32_t)((rs == QNRAW - 1 ? 0 : rs + 1) * QRAWB)); \
        if (qfin_ && qact) {                                                                                         \
            if ((I_) == 1) {                                                                                         \
                q_deq4(Qv[2], qd2, O[4], O[5]);                                                                      \
                G16Q_PIN2(O[4], O[5]);                                                                               \
            }                                                                                                        \
            if ((I_) == 2) {                                                                                         \
                q_deq4(Qv[3], qd2, O[6], O[7]);                                                                      \
                G16Q_PIN2(O[6], O[7]);                                                                               \
            }                                                                                                        \
            if ((I_) == 3) G16Q_WRITE(1 - qb_, 1);                                                                   \
        }                                                                                                            \
    } while (0)
            // second k-step: LDS-DMA of A stage kt + 4 (+ raw chunk kt / 2 + 3 when kt is even); the first 8 weights of stage kt + 2 into B slot qb_
#define G16Q_DQ_HOOK(I_)                                                                                              \
    do {                                                                                                             \
        if (qdma_ && (I_) < APW) q_issue_a((I_), kt0 + kt + NST, fbuf);                                              \
        if (qdma_ && qb_ == 0 && (I_) >= 2 && (I_) - 2 < QRQ) q_issue_raw_piece((I_) - 2 < QPI ? (I_) - 2 : QPI, (kt >> 1) + QNRAW, rs); \
        if (qconv_ && qact) {                                                                                        \
            if ((I_) == 0) {                                                                                         \
                q_extract1(qb_, Qv, qdbits);                                                                         \
                G16Q_PIN2(Qv[0], Qv[1]);                                                                             \
                G16Q_PIN2(Qv[2], Qv[3]);                                                                             \
                G16_TIE(qdbits);                                                                                     \
            }                                                                                                        \
            if ((I_) == 1) {                                                                                         \
                q_extract2(Qv, qdbits, qd2);                                                                         \
                G16Q_PIN2(Qv[0], Qv[1]);                                                                             \
                G16Q_PIN2(Qv[2], Qv[3]);                                                                             \
                G16_TIE(qd2);                                                                                        \
            }                                                                                                        \
            if ((I_) == 2) {                                                                                         \
                q_deq4(Qv[0], qd2, O[0], O[1]);                                                                      \
                G16Q_PIN2(O[0], O[1]);                                                                               \
            }                                                                                                        \
            if ((I_) == 3) {                                                                                         \
                q_deq4(Qv[1], qd2, O[2], O[3]);                                                                      \
                G16Q_PIN2(O[2], O[3]);                                                                               \
            }                                                                                                        \
            if ((I_) == 4) G16Q_WRITE(qb_, 0);                                                                       \
        }                                                                                                            \
    } while (0)
            static_assert(APW == 2 && QRQ <= 3, "hook slots: two A pieces, then up to three raw pieces");
            // DMA_: issue LDS-DMA (steady state); CONV_: start the conversion of stage kt + 2; FIN_: finish the conversion of stage kt + 1; W_: LDS-DMA instructions
            // that may stay in flight across the barrier
#define G16Q_ITER(B_, DMA_, CONV_, FIN_, W_)                                                                          \
    do {                                                                                                             \
        constexpr int qb_ = (B_);                                                                                    \
        constexpr bool qdma_ = (DMA_), qconv_ = (CONV_), qfin_ = (FIN_);                                             \
        const uint32_t sa = (uint32_t)buf * ABYTES;                                                                  \
        uint32_t O[8];                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        G16_TIE_FRAGS(A0, BH0);                                                                                      \
        G16_KSTEP_H(A0, A1, BH0, BH1, aad1 + sa, badq + (uint32_t)(qb_ * BBYTES), 1, G16Q_RD_HOOK);                   \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W_) : "memory");                                         \
        G16_TIE_FRAGS(A1, BH1);                                                                                      \
        if (qconv_) {                                                                                                \
            G16_TIE(RA);                                                                                             \
            G16_TIE(RB);                                                                                             \
            G16_TIE(RD);                                                                                             \
        }                                                                                                            \
        asm volatile("s_barrier" ::: "memory");                                                                      \
        const int fbuf = buf;                                                                                        \
        (void)fbuf;                                                                                                  \
        buf               = buf == NST - 1 ? 0 : buf + 1;                                                            \
        const uint32_t sn = (uint32_t)buf * ABYTES;                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        G16_KSTEP_H(A1, A0, BH1, BH0, aad0 + sn, badq + (uint32_t)((1 - qb_) * BBYTES), 0, G16Q_DQ_HOOK);             \
        ++kt;                                                                                                        \
    } while (0)
            // steady state, two iterations per pass (the block of the pair is a compile-time constant): kt even, kt + NST + 1 < nt
            for (; kt + NST < nt;) {
                G16Q_ITER(0, true, true, true, 2 * APW + QRQ);
                G16Q_ITER(1, true, true, true, 2 * APW + QRQ);
                rs = rs == QNRAW - 1 ? 0 : rs + 1;
            }
            // drain: stages nt - 4 .. nt - 1; the first two iterations still start the conversion of stages nt - 2 / nt - 1
            G16Q_ITER(0, false, true, true, 2 * APW + QRQ);
            G16Q_ITER(1, false, true, true, APW + QRQ);
            G16Q_ITER(0, false, false, true, 0);
            G16Q_ITER(1, false, false, false, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef G16Q_ITER
#undef G16Q_DQ_HOOK
#undef G16Q_PIN2
#undef G16Q_RD_HOOK
#undef G16Q_WRITE
#undef G16Q_RAW_READ
        } else {
        if (WEXTRA != 0 && w_short)
            G16_PIPE_LOOP(NPT - 1);
        else
            G16_PIPE_LOOP(NPT);
        }
#undef G16_PIPE_LOOP
#undef G16_DMA_HOOK
#undef G16_TIE_FRAGS
#undef G16_KSTEP_H
#undef G16_KSTEP
#undef G16_NOHOOK
#undef G16_TIE
#undef G16_RD
    } else if (NST == 2) {
        // one barrier per K tile: its implicit vmcnt(0) retires this tile's DMA, the next tile's DMA overlaps the MFMAs
        stage(kt0, 0, c_tap, c_kh, c_kw, c_icb, c_sub);
        G16_ADVANCE();
        for (int kt = 0; kt < nt; ++kt) {
            __syncthreads();
            if (kt + 1 < nt) {
                stage(kt0 + kt + 1, (kt + 1) & 1, c_tap, c_kh, c_kw, c_icb, c_sub);
                G16_ADVANCE();
            }
            compute(kt & 1);
        }
    } else {
        // NST-deep ring, COUNTED waits: NST-1 tiles of DMA are issued ahead and up to NST-2 stay in flight across the barrier
        // (cdna_hip_programming.md T3+T4): the wait retires tile kt while the younger tiles keep streaming; tile kt+NST-1 is issued right
        // after the barrier into the slot tile kt-1 just left.  (Deeper rings were measured for launches that leave one workgroup per CU —
        // 5 and 6 stages on the 128x128 tile, profiles/r02q_tile_sweep.txt — and change nothing: those k-steps are not bound by the
        // prefetch distance.)
        constexpr int PD = NST - 1;
        static_assert(PD * NPT <= 63, "vmcnt is a 6-bit counter");
        for (int i = 0; i < PD && i < nt; ++i) {
            stage(kt0 + i, i, c_tap, c_kh, c_kw, c_icb, c_sub);
            G16_ADVANCE();
        }
        int buf = 0, fill = PD % NST;  // fill = slot of the next tile to stage
        for (int kt = 0; kt < nt; ++kt) {
            const int ahead = min(PD - 1, nt - 1 - kt);  // younger tiles that may stay in flight
            if (ahead <= 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (w_short) {
                switch (ahead) {
                    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * (NPT - 1)) : "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NPT - 1)) : "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NPT - 1)) : "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * (NPT - 1)) : "memory"); break;
                }
            } else {
                switch (ahead) {
                    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * NPT) : "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPT > 63 ? 63 : 2 * NPT) : "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPT > 63 ? 63 : 3 * NPT) : "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NPT) : "memory"); break;
                }
            }
            __builtin_amdgcn_s_barrier();
            if (kt + PD < nt) {
                stage(kt0 + kt + PD, fill, c_tap, c_kh, c_kw, c_icb, c_sub);
                G16_ADVANCE();
                fill = fill == NST - 1 ? 0 : fill + 1;
            }
            compute(buf);
            buf = buf == NST - 1 ? 0 : buf + 1;
        }
    }

    // ---- split-K, reduced in the launch: slab dump -> ticket; the last arriver sums the slices in slice order (bitwise deterministic whoever
    // arrives last) and falls through to the epilogue.  Protocol = cdna_hip_programming.md section 5 "in-launch split-K reduction", write-through
    // form: 16-byte sc1 stores, every storing wave drains vmcnt, barrier, ONE lane takes a relaxed agent-scope ticket; the reducer reads with
    // sc1 loads.  Nobody ever waits for another workgroup, so residency does not matter.  The plan zeroes all its tile counters with one hipMemsetAsync ahead of its first launch.
    // Only the 128-row tiles take this path (64 / 32 KB per slice: a few slices are a few microseconds of reading for the last arriver; the
    // 320 KB slabs of the 256x320 tile were measured 40-70 % slower than slabs + a chip-wide reduce pass, profiles/r02o_*).
    if (BM == 128 && g.split_k > 1 && g.sk_cnt) {
        // slabs are stored WRITE-THROUGH (sc1, aux = 16) and read back with sc1 loads, so the hand-off needs no release / acquire fence: a
        // buffer_wbl2 per workgroup writes back every dirty line of the XCD's L2 — with all workgroups dumping slabs at once that cost
        // ~20 us per launch (r02q: SDXL Linears 23 -> 36 ms with fences)
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        constexpr int TILE_B = BM * BN * 4;
        char* slab0          = (char*)g.sk_slab + (int64_t)bid * g.split_k * TILE_B;  // wave-uniform
        const auto rs        = __builtin_amdgcn_make_buffer_rsrc((void*)slab0, 0, g.split_k * TILE_B, 0x00020000);
        const int mine       = kslice * TILE_B;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    u32x4_t v;
                    v[0] = __float_as_uint(acc[rb][cb][4 * i4]);
                    v[1] = __float_as_uint(acc[rb][cb][4 * i4 + 1]);
                    v[2] = __float_as_uint(acc[rb][cb][4 * i4 + 2]);
                    v[3] = __float_as_uint(acc[rb][cb][4 * i4 + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine + ((((wave * RB + rb) * CB + cb) * 4 + i4) * 64 + lane) * 16, 0, 16);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
        __syncthreads();
        int* flag = (int*)smem;  // the staging ring is dead: every wave is past its last fragment read
        if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(&g.sk_cnt[bid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.split_k - 1;
        __syncthreads();
        if (!*flag) return;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = (float16_t){0};
        for (int sl = 0; sl < g.split_k; ++sl) {  // slice order: the sum does not depend on who arrives last
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, sl * TILE_B + ((((wave * RB + rb) * CB + cb) * 4 + i4) * 64 + lane) * 16, 0, 16);
                        acc[rb][cb][4 * i4] += __uint_as_float(v[0]);
                        acc[rb][cb][4 * i4 + 1] += __uint_as_float(v[1]);
                        acc[rb][cb][4 * i4 + 2] += __uint_as_float(v[2]);
                        acc[rb][cb][4 * i4 + 3] += __uint_as_float(v[3]);
                    }
        }
    }

    // ---- stream-K: a segment that is not a whole tile is one PART of its tile
    if constexpr (SK) {
        __syncthreads();  // every wave is past its last fragment read: the ring is dead (flag below) and may be restaged by the next segment
        if (!(kt0 == 0 && nt == g.nt)) {
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            constexpr int TILE_B = BM * BN * 4;
            // parts of tile `bid` in K order: part p is computed by (logical) workgroup wf + p, wf = owner of the tile's first unit; owner of unit u =
            // floor(((u + 1) * G - 1) / U).  Slab slots: two per workgroup — [2w] for its segment with kt0 > 0 (at most one: its first), [2w + 1] for
            // its segment starting at kt0 == 0 and cut short by the end of its range (at most one: its last)
            const int64_t U  = (int64_t)(g.sk_tiles - g.sk_dp) * g.nt, G = g.sk_grid;
            const int64_t uf = (int64_t)(bid - g.sk_dp) * g.nt;
            const int wf     = (int)(((uf + 1) * G - 1) / U), wl = (int)(((uf + g.nt) * G - 1) / U);
            const int nparts = wl - wf + 1;
            char* mine       = (char*)g.sk_slab + (int64_t)(2 * sk_w + (kt0 == 0 ? 1 : 0)) * TILE_B;  // wave-uniform
            {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, TILE_B, 0x00020000);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            u32x4_t v;
                            v[0] = __float_as_uint(acc[rb][cb][4 * i4]);
                            v[1] = __float_as_uint(acc[rb][cb][4 * i4 + 1]);
                            v[2] = __float_as_uint(acc[rb][cb][4 * i4 + 2]);
                            v[3] = __float_as_uint(acc[rb][cb][4 * i4 + 3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((((wave * RB + rb) * CB + cb) * 4 + i4) * 64 + lane) * 16, 0, 16);  // sc1: write-through
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
            __syncthreads();
            int* flag = (int*)smem;
            if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(&g.sk_cnt[bid - g.sk_dp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nparts - 1;
            __syncthreads();
            const int last = *flag;
            __syncthreads();  // the flag word belongs to the ring the next segment stages into
            if (!last) continue;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = (float16_t){0};
            for (int p = 0; p < nparts; ++p) {  // part order: the sum does not depend on who arrives last
                const char* part = (const char*)g.sk_slab + (int64_t)(p == 0 ? 2 * wf + 1 : 2 * (wf + p)) * TILE_B;
                const auto rp    = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, TILE_B, 0x00020000);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rp, ((((wave * RB + rb) * CB + cb) * 4 + i4) * 64 + lane) * 16, 0, 16);
                            acc[rb][cb][4 * i4] += __uint_as_float(v[0]);
                            acc[rb][cb][4 * i4 + 1] += __uint_as_float(v[1]);
                            acc[rb][cb][4 * i4 + 2] += __uint_as_float(v[2]);
                            acc[rb][cb][4 * i4 + 3] += __uint_as_float(v[3]);
                        }
            }
        }
    }

    // ---- epilogue: one compact variant per workgroup (all conditions are launch- or workgroup-uniform)
    if constexpr (SWP) {
        epi_dispatch_linear_swp(acc, g, row0, col0, wr, wc, lane);
    } else if (!CONV) {
        // column-range epilogue (Epilogue::split_col): this tile belongs to the f32 output or to the f16 gelu tensor.  Compiled into the pipelined
        // 256 x 256 tile only (the epilogue code exists twice there): gemm16_split_col_supported tells the planner which launches take that tile
        if (PIPE == 1 && BN == 256 && g.split_col > 0) {
            const bool hi_part = col0 >= g.split_col;
            G16Args gs   = g;  // (a workgroup-uniform copy: the stream-K loop must see the launch's own fields again in its next pass)
            gs.dst       = hi_part ? nullptr : g.dst;
            gs.dst16     = hi_part ? g.split_dst16 : nullptr;
            gs.ldd16     = hi_part ? g.split_ldd16 : 0;
            gs.ep.gelu   = hi_part ? 1 : 0;
            epi_dispatch_linear<BM>(acc, gs, row0, col0, wr, wc, lane);
        } else
        epi_dispatch_linear<BM>(acc, g, row0, col0, wr, wc, lane);
    } else {
        const bool fullc = col0 + BN <= g.C;
        if (fullc) {
            if (g.ep.residual)
                epi_conv<1>(acc, g, row0, col0, wr, wc, lane);
            else
                epi_conv<0>(acc, g, row0, col0, wr, wc, lane);
        } else {
            epi_conv<2>(acc, g, row0, col0, wr, wc, lane);
        }
    }
  }  // tile segments
}

static inline int64_t rup64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// pipeline / tile variant (tunable at run time for A/B measurements):
//   0 = BK 64 x 2 stages (__syncthreads), 1 = BK 32 x 3 stages (counted vmcnt), 2 = BK 64 x 3 stages; all 128-row tiles
//   3 (default) = variant 1 plus 256-row x 128-col tiles (8 waves) whenever they still fill the chip — the LDS-DMA engine
//   delivers ~20 B/clk/CU, so FLOP per DMA byte (tile area / perimeter) is what bounds these kernels.
static int g_g16_variant = 3;
static int g_g16_tap_major = 0;
void gemm16_set_tap_major(int v) { g_g16_tap_major = v; }
int gemm16_tap_major() { return g_g16_tap_major; }
void gemm16_set_variant(int v) { g_g16_variant = v; }
static inline bool g16_bk32() { return g_g16_variant == 1 || g_g16_variant == 3; }

// tile configurations (all BK 32 x 3 stages, counted vmcnt):
//   T128   128x128, 4 waves of 64x64   (48 KB LDS, 3 workgroups/CU)  — small outputs, finest granularity
//   T256   256x128, 8 waves of 64x64   (72 KB, 2/CU)                 — large outputs, 1.33x the per-round area of T128
//   T256W  256x128, 4 waves of 128x64  (72 KB, 2/CU)                 — same tile, fatter waves: 25 % fewer LDS fragment reads per MFMA
//   T160   256x160, 4 waves of 64x160  (78 KB, 2/CU)                 — outputs that are multiples of 160 but not of 128 (SD1.5's 320):
//                                                                       no padded columns, 2 column tiles instead of 3
//   T160N  256x160, 8 waves of 32x160  (78 KB, 2/CU)                 — T160 with twice the waves in flight (experiment)
enum { G16_T128 = 0, G16_T256 = 1, G16_T256W = 2, G16_T160 = 3, G16_T160N = 4, G16_T320 = 5, G16_T256P = 6, G16_T128N64 = 7, G16_T192P = 8 };  // T192P: the pipelined loop on 256 x 192 tiles (Linears only)
static int g_g16_force_tile = -1;  // option "gemm16_tile": force one configuration (A/B measurements); -1 = choose per shape
void gemm16_set_tile(int t) { g_g16_force_tile = t; }
// Per-shape choice.  Measured on SD1.5 batch 16 (profiles/r01e_tile_configs.txt): a launch takes ceil(workgroups / resident slots)
// rounds; a full round of T128 (768 slots) and of T256 (512 slots, twice the area per workgroup) take about the same time, a T160
// round 1.5x that (4 waves per workgroup hide less latency) but covers 1.25x T256's area with no padded columns.  256-row tiles
// only pay when they fill every CU twice (>= 512 workgroups); otherwise the finer T128 quantises better.
#ifdef MI355X_EXPERIMENTS
static int g_g16_abl = 0;  // option "gemm16_abl": 1 = no MFMAs, 2 = no DMA after the fill, 3 = conv input tiles fetched for tap 0 only (T320 only; wrong results, timing)
void gemm16_set_abl(int v) { g_g16_abl = v; }
#endif
static int g_g16_bn64 = 1;  // option "gemm16_bn64": 0 = 64-column tiles only for M <= 64 (A/B measurements)
void gemm16_set_bn64(int v) { g_g16_bn64 = v; }
static int g_g16_streamk = 0;  // option "streamk" (g16_streamk_grid below): 0 (default) = off — measured neutral to slightly negative on the FLUX forward even restricted to launches of two rounds or more (profiles/r05e_family_flux_split_gelu_streamk.txt); 1 = that policy; 2 = every candidate
static int g_g16_t256p_min_nt_sk = 64, g_g16_t256p_min_tiles_sk = 192;  // options "t256p_min_nt_sk" / "t256p_min_tiles_sk": K stages / tiles from which a stream-K-able Linear takes the 256 x 256 tile
void gemm16_set_t256p_min_nt_sk(int v) { g_g16_t256p_min_nt_sk = v; }
void gemm16_set_t256p_min_tiles_sk(int v) { g_g16_t256p_min_tiles_sk = v; }
static int g_g16_t320 = 1;  // option "gemm16_t320": 0 disables the pipelined 256x320 tile in the per-shape choice (A/B measurements)
void gemm16_set_t320(int v) { g_g16_t320 = v; }
int gemm16_split_k(int64_t rows, int64_t M, int64_t K, bool conv);
static int g16_t320_split(int64_t rows, int64_t M, int64_t nt, bool conv);
// mul > 1: `mul` sibling weights of M columns each in one launch (divisibility per weight, tile counts over all of them)
// geglu: 0 = plain Linear, 1 = GEGLU launch on the paired weight image (128-column pairing: tiles with an even number of column blocks per wave
// only), 2 = GEGLU launch on the 16-column interleave (epi_geglu16: any tile)
// option "t256p_pad" (round 5): the pipelined 256 x 256 tile also for Linears whose width is a multiple of 128 but not of 256 (SD3.5-large: 2432 = 9.5 tiles,
// 7296 = 28.5): the last column tile is half empty (its weight fetches are clamped to the image, its outputs masked), at most 6 % of the launch
static int g_g16_t192p = 1;  // option "gemm16_t192p": 0 = never take the 256 x 192 tile by itself
void gemm16_set_t192p(int v) { g_g16_t192p = v; }
static int g_g16_t256p_pad = 1;
void gemm16_set_t256p_pad(int v) { g_g16_t256p_pad = v; }
// option "tail_split" (round 5): a 256 x 256-tile Linear whose tile count leaves the last of its >= 2 rounds mostly empty runs as TWO launches split by rows:
// whole rounds of 256 x 256 tiles, then the remaining rows on the small tiles (g16_tail_rows).  No slab traffic, no reduction — unlike stream-K.
static int g_g16_tail_split = 0;  // default OFF: measured neutral on FLUX (57.64 vs 57.67 ms of Linear kernels per forward) and SD3.5 (58.23 vs 58.17), profiles/r06b_ab_*_tail_split.txt
void gemm16_set_tail_split(int v) { g_g16_tail_split = v; }
static bool g16_pad256_ok(int64_t M, int geglu, bool conv, int mul) {
    return g_g16_t256p_pad && !conv && mul == 1 && geglu == 0 && M % 128 == 0 && M % 256 != 0 && M >= 2048;  // >= 2048: the empty half tile is <= 6 % of the columns
}
// row tiles (256 rows) the main launch of a row-split Linear takes; 0 = one launch.  Unit of the estimate: one round of 256 x 256 tiles on 256 CUs
static int g16_num_cus();
static int g16_tail_rows(int64_t rows, int64_t M) {
    if (!g_g16_tail_split) return 0;
    const int64_t ncol = (M + 255) / 256, rt = (rows + 255) / 256, T = rt * ncol, full = T / g16_num_cus(), rem = T % g16_num_cus();
    if (full < 1 || rem == 0) return 0;
    const int64_t rtm = full * g16_num_cus() / ncol;
    if (rtm <= 0 || rtm >= rt) return 0;
    const int64_t tail_rows = rows - rtm * 256, tail_wgs = ((tail_rows + 127) / 128) * ((M + 127) / 128);
    // tail on 128 x 128 tiles (3 workgroups per CU at most): one per CU ~ half a big tile's time, two ~ 0.75, a full round of three ~ 1.25
    const double tail_cost = tail_wgs <= 256 ? 0.5 : (tail_wgs <= 512 ? 0.75 : 1.25 * (double)((tail_wgs + 767) / 768));
    const double main_cost = (double)((rtm * ncol + g16_num_cus() - 1) / g16_num_cus());
    return (main_cost + tail_cost < (double)(full + 1) * 0.95) ? (int)rtm : 0;
}
static int g16_pick_tile(int64_t rows, int64_t M, int geglu, bool conv, int split, int64_t nt, int mul = 1) {
    if (g_g16_variant != 3) return G16_T128;
    const bool can160 = M % 160 == 0 && geglu != 1;
    // T320 (256x320, one workgroup per CU): the weight image is padded to 128 columns only, so M must be a multiple of 320; the GEGLU
    // pairing is laid out for 128-column tiles
    const bool can320 = M % 320 == 0 && geglu != 1;
    if (g_g16_force_tile >= 0) {
        if (g_g16_force_tile == G16_T320) return can320 ? G16_T320 : G16_T256;
        if (g_g16_force_tile == G16_T256P) return M % 256 == 0 ? G16_T256P : G16_T256;
        if (g_g16_force_tile == G16_T192P) return ((M % 192 == 0 || (M % 64 == 0 && M >= 2048)) && !conv && geglu == 0 && !split) ? G16_T192P : G16_T256;
        if ((g_g16_force_tile == G16_T160 || g_g16_force_tile == G16_T160N) && !can160) return G16_T256;
        return g_g16_force_tile > G16_T256P ? G16_T128 : g_g16_force_tile;
    }
    const int64_t rt256 = (rows + 255) / 256, c128 = ((rows + 127) / 128) * ((M + 127) / 128) * mul, c256 = rt256 * ((M + 127) / 128) * mul;
    if (g_g16_force_tile < 0 && !split && g_g16_t320 && !can320 && (M % 256 == 0 || g16_pad256_ok(M, geglu, conv, mul))) {
        // T256P: the same pipelined loop on 256x256 tiles (N a multiple of 256 but not of 320: DiT Linears, the KL-VAE's 512 / 256-channel convs)
        // one workgroup per CU: pipeline fill, drain and epilogue of a workgroup overlap with nothing, so short-K GEMMs (SD1.5's GEGLU FF1,
        // K = 320 .. 1280: 10-40 stages) stay on the 2-workgroups-per-CU tiles (r02d: 264 -> 318 us); long-K Linears (DiT) take it
        const int64_t c256p = rt256 * ((M + 255) / 256) * mul;
        int64_t rounds      = (c256p + 255) / 256;
        if (!conv && mul == 1 && geglu == 0 && g16_tail_rows(rows, M) > 0) rounds = c256p / 256;  // the partial round goes to the tail launch: the fill test sees whole rounds
        // stream-K (g16_streamk_grid) runs such a launch as ONE round whatever its tile count: the round-fill test only binds launches that cannot take it
        const bool sk_ok = g_g16_streamk == 2 && !conv && mul == 1 && c256p * nt >= 256 * 16;  // (default policy: stream-K never widens the tile choice)
        const bool ok256 = nt >= (sk_ok ? g_g16_t256p_min_nt_sk : 64) && c256p >= (sk_ok ? g_g16_t256p_min_tiles_sk : 192) && (sk_ok || c256p * 4 >= rounds * 256 * 3);
        // 256 x 192 tiles (round 6) where they quantise better on the chip: FLUX's 4096-row img stream has 192 (-> 3072) / 576 (-> 9216) tiles of 256 x 256 = one / three
        // rounds on 256 CUs of which a quarter is idle; 256 / 768 tiles of 256 x 192 fill them.  A round of the narrower tile costs ~0.8 of a 256 x 256 round
        // (gpurun_out/r08g: 88.7 -> 80.6, 309.7 -> 280.5, 259.0 -> 228.3 us; equal round counts lose: -> 12288 292.6 vs 299.7 us).
        // Widths that are not multiples of 192 (SD3.5-large: 2432 = 12.67 tiles, 9728 = 50.67) take it like the padded 256 x 256 tile does: the last column tile's weight
        // fetches are clamped to the image, its outputs masked — at most 6 % of the launch from 2048 columns on.  Same fill rule as the 256 x 256 tile (>= 192 tiles,
        // rounds at least 3/4 full); where only the narrower tile passes it (SD3.5: 8500 x 2432 = 340 tiles of 256 x 256 = 1.33 rounds, 442 of 256 x 192 = 1.73) it is taken
        if (g_g16_t192p && !conv && mul == 1 && geglu == 0 && (M % 192 == 0 || (g_g16_t256p_pad && M % 64 == 0 && M >= 2048)) && !sk_ok && nt >= 64) {
            const int64_t c192 = rt256 * ((M + 191) / 192), r192 = (c192 + 255) / 256;
            const bool ok192   = c192 >= 192 && c192 * 4 >= r192 * 256 * 3;
            if (ok192 && (!ok256 || (double)r192 * 0.80 < (double)rounds * 0.97)) return G16_T192P;
        }
        if (ok256) return G16_T256P;
    }
    if (split) {
        if (can320 && g16_t320_split(rows, M, nt, conv) == split) return G16_T320;
    } else if (g_g16_t320 && can320) {
        // one workgroup per CU and 256 CUs: take it when the launch fills >= 75 % of its rounds
        const int64_t c320 = rt256 * (M / 320) * mul, rounds = (c320 + 255) / 256;
        // GEGLU launches (16-column interleave): a single round only — the heavier epilogue of a one-workgroup-per-CU tile overlaps with nothing, and from two
        // rounds on the 256 x 128 tile (two workgroups per CU) wins: 4096 x 1280 -> 10240 (512 tiles) 187.9 vs 191.5 us for FF1 + FF2, against 121.7 ->
        // 113.0 us at 2048 rows (256 tiles = SDXL's 32 x 32 level), profiles/r05f_ff_probe_geglu16.txt
        if (nt >= (conv ? 16 : 32) && c320 >= 192 && c320 * 4 >= rounds * 256 * 3 && (geglu != 2 || rounds == 1)) return G16_T320;
    }
    const int64_t c160 = can160 ? rt256 * (M / 160) * mul : 0;
    double best = (double)((c128 + 767) / 768) * 1.0;
    int tile    = G16_T128;
    // a partially filled T256 round (256..511 workgroups: some CUs host two, most one) runs ~1.2x a full round's time per workgroup
    const double cost256 = (double)((c256 + 511) / 512) * 0.98 * (c256 < 512 ? 1.2 : 1.0);
    if (c256 >= 256 && cost256 < best) {
        best = cost256;
        tile = G16_T256;
    }
    if (c160 >= 512 && (double)((c160 + 511) / 512) * 1.52 < best) tile = conv ? G16_T160 : G16_T160N;  // short-K linears: 8 thin waves hide more latency
    else if (tile == G16_T128 && c160 >= 256 && c160 < 512 && c128 <= 768) tile = G16_T160N;             // one workgroup per CU, 8 waves each: ~4 % over T128
    return tile;
}

bool gemm16_split_col_supported(int64_t rows, int64_t M, int64_t K);
int gemm16_geglu_mode(int64_t rows, int64_t M, int64_t K);
static bool g16_use_bn64(int64_t rows, int64_t M, int mul);
static int g_g16_swp = 0;  // option "gemm16_swp": 1 = big-token Linear tiles with the accumulator transposed (16-byte epilogue accesses; measured slower: profiles/r05a_ab_gemm16_swp_rejected.txt)
// ---- stream-K policy (option "streamk", default 0 = off: measured 2.8 % slower, profiles/r05i_*).  A Linear that g16_pick_tile sends to a one-workgroup-per-CU pipelined tile (T320 / T256P) and
// whose tile count leaves the last round mostly empty runs as ONE round of persistent workgroups over equal (tile, K-tile) unit ranges instead
// (k_gemm16<..., SK>).  Returns the grid (= CUs), or 0.  rows / M / K of ONE weight; not for grouped (multi) launches, GEGLU, or K-split launches.
void gemm16_set_streamk(int v) { g_g16_streamk = v; }
static int g16_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}
static int g16_streamk_grid(int64_t rows, int64_t M, int64_t K, bool geglu, int* tiles_out = nullptr, int* bn_out = nullptr, int* dp_out = nullptr) {
    if (!g_g16_streamk || g_g16_variant != 3 || !g16_bk32() || g_g16_force_tile >= 0 || g_g16_swp) return 0;
    const int64_t nt = rup64(K, 64) / 32;
    if (!geglu && gemm16_split_k(rows, M, K, false) > 1) return 0;
    const int tile = g16_pick_tile(rows, M, geglu ? gemm16_geglu_mode(rows, M, K) : 0, false, 0, nt, 1);
    // (the 256 x 320 tile's stream-K instantiation needs 21 spilled registers on top of its 160 accumulators; its Linears — M a multiple of 320: the
    // UNets — have tile counts that fill their rounds or take K slices, so only the 256 x 256 tile (DiT widths) is instantiated)
    if (tile != G16_T256P) return 0;
    const int bn        = 256;
    const int64_t tiles = ((rows + 255) / 256) * ((M + bn - 1) / bn);
    const int cus       = g16_num_cus();
    const int64_t rounds = (tiles + cus - 1) / cus;
    // Measured (FLUX.1-dev shapes, profiles/r05c_streamk_*.txt): every range boundary cuts a tile, so all workgroups dump and combine 256 KB slabs at the SAME
    // time at the end of their ranges.  With at least two full rounds of work behind it that costs less than the empty part of the last round when
    // that part is large (4096 x 3072 -> 9216: 576 tiles = 2.25 rounds, 271 -> 255 us); it loses when the last round is nearly full (1428 tiles = 5.58
    // rounds: +1.5 %) and loses badly when the whole launch is one partial round, where EVERY tile is cut (192 tiles: 123 -> 153 us, 204 tiles,
    // K = 15360: 412 -> 457 us).  Hence: two rounds or more, and at least 15 % of the rounds empty (option "streamk" = 2: every candidate, for A/B runs).
    if (tiles * nt < (int64_t)cus * 16 || tiles >= (1 << 20)) return 0;
    if (g_g16_streamk == 1 && (rounds < 2 || tiles * 100 >= rounds * cus * 85)) return 0;
    if (g_g16_streamk == 2 && tiles * 100 >= rounds * cus * 92) return 0;
    // 3 = HYBRID: the full rounds stay whole tiles (sk_dp of them), only the tiles of the partial last round are cut — over all workgroups, before anything else.
    // Worth it when that round is at most 3/4 full and a workgroup's share of it is still a pipeline's worth of K stages
    int64_t dp = 0;
    if (g_g16_streamk == 3) {
        dp = (tiles / cus) * cus;
        const int64_t rem = tiles - dp;
        if (dp == 0 || rem == 0 || rem * 4 > (int64_t)cus * 3 || rem * nt < (int64_t)cus * 12) return 0;
    }
    if (dp_out) *dp_out = (int)dp;
    if (tiles_out) *tiles_out = (int)tiles;
    if (bn_out) *bn_out = bn;
    return cus;
}

void gemm16_set_swp(int v) { g_g16_swp = v; }
// can this launch's epilogue run on the transposed accumulator?  (f32 +bias +residual | f16 rows (+GELU) | GEGLU; whole 32-column blocks; vector alignment)
static bool g16_swp_ok(const G16Args& g) {
    if (!g_g16_swp || g.split_k > 1 || g.multi > 1 || g.hm_d > 0 || g.ep.gate || g.C % 32 != 0) return false;
    auto al = [](const void* p, int a) { return (((uintptr_t)p) & (uintptr_t)(a - 1)) == 0; };
    if (g.ep.bias && !al(g.ep.bias, 16)) return false;
    if (g.geglu_inner > 0) return g.dst16 && !g.dst && !g.ep.residual && g.geglu_inner % 32 == 0 && g.ldd16 % 4 == 0 && al(g.dst16, 8);
    if (g.dst16) return !g.dst && !g.ep.residual && g.ldd16 % 4 == 0 && al(g.dst16, 8);
    return g.dst && !g.ep.gelu && g.ldd % 4 == 0 && al(g.dst, 16) && (!g.ep.residual || al(g.ep.residual, 16));
}

// weight image layout of a GEGLU FF1 of this shape (M = 2 * inner columns): 2 = 16-column interleave (the launch takes a tile with an odd number of
// column blocks per wave: 256 x 160 / 256 x 320), 1 = 128-column pairing.  Option "geglu16" (1): 0 = always pairing (round-3 behaviour)
static int g_g16_geglu16 = 1;
void gemm16_set_geglu16(int v) { g_g16_geglu16 = v; }
int gemm16_geglu_mode(int64_t rows, int64_t M, int64_t K) {
    if (!g_g16_geglu16 || g_g16_variant != 3 || !g16_bk32() || g_g16_force_tile >= 0 || (M / 2) % 16 != 0) return 1;
    const int tile = g16_pick_tile(rows, M, 2, false, 0, rup64(K, 64) / 32, 1);
    return (tile == G16_T160 || tile == G16_T160N || tile == G16_T320) ? 2 : 1;
}
// in-loop dequantisation (k_gemm16<..., QT>): shapes whose launch takes the pipelined 256 x 256 tile anyway, whole column tiles, an even number (>= 6) of K stages
// per K slice, weight rows that start 4-byte aligned (K / 32 even: K % 64 == 0).  Option "qinloop_min_rows" (default 513 = right above k_qgemm16's range; 0 = off)
static int g_g16_qinloop_min_rows = 513;
void gemm16_set_qinloop_min_rows(int v) { g_g16_qinloop_min_rows = v; }
bool gemm16_qinloop_supported(int wtype, int64_t rows, int64_t M, int64_t K, int mul, int split) {
    if (g_g16_qinloop_min_rows <= 0 || rows < g_g16_qinloop_min_rows || (wtype != 8 && wtype != 2)) return false;
    if (g_g16_variant != 3 || !g16_bk32() || g_g16_swp || g_g16_force_tile >= 0 || g_g16_streamk) return false;
    if (K % 64 != 0 || K < 192 || g16_use_bn64(rows, M, mul)) return false;
    const int64_t nt = K / 32;
    if (split > 1 || gemm16_split_k(rows, M, K, false) > 1) return false;  // (K slices: not yet)
    const int tile = g16_pick_tile(rows, M, 0, false, 0, nt, mul);
    return (tile == G16_T256P || tile == G16_T192P) && M % 64 == 0;  // (ragged last column tiles re-fetch the last weight row)
}
// a Linear of this shape runs on the pipelined 256 x 256 tile without K slices: the launches that may carry Epilogue::split_col
bool gemm16_split_col_supported(int64_t rows, int64_t M, int64_t K) {
    if (g_g16_variant != 3 || !g16_bk32() || g_g16_swp || M % 256 != 0 || M % 160 == 0 || g16_use_bn64(rows, M, 1)) return false;
    const int64_t nt = rup64(K, 64) / 32;
    return gemm16_split_k(rows, M, K, false) == 1 && g16_pick_tile(rows, M, false, false, 0, nt, 1) == G16_T256P;
}

// option "conv_wmajor" (round 5, default 1): conv launches whose weight image is at least twice their NHWC input image run in weight-major workgroup order
// (G16Args::worder, g16_wg_order).  0 = the A-major order everywhere (A/B measurements).
static int g_g16_conv_wmajor = 1;
void gemm16_set_conv_wmajor(int v) { g_g16_conv_wmajor = v; }
int gemm16_worder_rows(const G16Args& g, unsigned gx, unsigned ny) {
    if (!g_g16_conv_wmajor || g.KS == 0 || g.ncol_tiles <= 0 || g.multi > 1 || g.sk_grid > 0) return 0;
    if (gx % 8 != 0 || gx % (unsigned)g.ncol_tiles != 0 || ((unsigned)g.ncol_tiles * ny) % 8 != 0) return 0;
    const int nrow = (int)(gx / (unsigned)g.ncol_tiles);
    if (nrow < 2) return 0;  // one row tile: nothing shares a weight chunk
    const int64_t imgs = g.OHOW > 0 ? g.R / g.OHOW : 1;
    const double act   = (double)imgs * g.H * g.Wd * g.ICp * 2.0;
    const double wts   = (double)g.ICp * g.KS * g.KS * (double)rup64(g.C, 128) * 2.0;
    return wts >= 2.0 * act ? nrow : 0;
}
static bool g16_trace();
template <int BN_, bool CONV_>
static void g16_launch(hipStream_t s, G16Args& g, int64_t rows, double flops, double bytes) {  // bytes: algorithmic HBM bytes (operand images read once + output written once [+ residual])
    const unsigned ny = g.split_k > 1 ? (unsigned)g.split_k : 1u;
    const int mul     = (!CONV_ && g.multi > 1) ? g.multi : 1;  // sibling Linears in one launch: mul x the column tiles
    if (BN_ == 128 && g_g16_variant == 3 && (!g.sk_cnt || g.sk_grid > 0)) {
        const int tile = g16_pick_tile(rows, g.C, g.geglu_inner > 0 ? (g.geglu16 ? 2 : 1) : 0, CONV_, g.split_k > 1 ? g.split_k : 0, g.nt, mul);  // the GEGLU pairing is laid out for 128-column tiles
        if (g16_trace()) fprintf(stderr, "G16 tile %d rows=%lld C=%lld nt=%d split=%d mul=%d qt=%d\n", tile, (long long)rows, (long long)g.C, g.nt, g.split_k, mul, g.qt);
        if (g.qt && ((tile != G16_T256P && tile != G16_T192P) || g.sk_grid > 0 || g.C % 64 != 0 || g.nt < 6 || (g.nt & 1) || (g.split_k > 1 && ((g.nt_slice & 1) || g.nt_slice < 6 || g.nt - (g.split_k - 1) * g.nt_slice < 6)))) {
            fprintf(stderr, "ggml-mi355x: in-loop dequantisation planned for a launch that does not take the pipelined 256 x 256 tile (tile %d, rows %lld, M %lld, K stages %d)\n", tile, (long long)rows, (long long)g.C, g.nt);
            abort();
        }
        if (g.geglu_inner > 0 && g.geglu16 && tile != G16_T160 && tile != G16_T160N && tile != G16_T320) {
            fprintf(stderr, "ggml-mi355x: GEGLU launch planned on the 16-column interleave, but its tile takes the paired image\n");
            abort();
        }
        if constexpr (!CONV_) {
            if (g.sk_grid > 0) {  // stream-K: one persistent workgroup per CU (the planner asked g16_streamk_grid, which made the same tile choice)
                KScope ks_(s, KF_LINEAR, flops, bytes);
                if (tile == G16_T256P) {
                    g.ncol_tiles = (int)((g.C + 255) / 256);
                    k_gemm16<256, 256, false, 32, 4, 4, 2, 1, false, true><<<dim3((unsigned)g.sk_grid, 1), 512, 0, s>>>(g);
                } else {
                    fprintf(stderr, "ggml-mi355x: stream-K launch planned for a shape that does not take a pipelined tile\n");
                    abort();
                }
                return;
            }
        }
        if constexpr (!CONV_) {
            const int rtm = (tile == G16_T256P && ny == 1 && mul == 1 && g.geglu_inner == 0 && !g.sk_cnt && g.split_col == 0 && !g.qt) ? g16_tail_rows(rows, g.C) : 0;  // the tail runs on tiles whose epilogue has no column-range store (split_col is compiled into the 256 x 256 pipelined tile only)
            if (rtm > 0) {
                // row split: whole rounds of 256 x 256 tiles, then the remaining rows on whatever tile their shape picks (epilogue indices are absolute rows)
                const double fm = (double)((int64_t)rtm * 256) / (double)rows;
                G16Args t       = g;
                t.row_base      = g.row_base + (int64_t)rtm * 256;
                {
                    KScope ks_(s, KF_LINEAR, flops * fm, bytes * fm);
                    g.ncol_tiles = (int)((g.C + 255) / 256);
                    if (g.C % 256 != 0) g.wblk_lim = (int)(rup64(g.C, 128) / 32);
                    k_gemm16<256, 256, false, 32, 4, 4, 2, 1><<<dim3((unsigned)(rtm * g.ncol_tiles), 1), 512, 0, s>>>(g);
                }
                g16_launch<BN_, CONV_>(s, t, rows - (int64_t)rtm * 256, flops * (1.0 - fm), bytes * (1.0 - fm));
                return;
            }
        }
        if (tile != G16_T128) {
            const int64_t rt256 = (rows + 255) / 256;
            KScope ks_(s, CONV_ ? KF_CONV_T256 : KF_LINEAR, flops, bytes);
            if constexpr (CONV_) {  // weight-heavy convs (8x8 / 16x16 UNet levels): weight-major workgroup order
                G16Args t    = g;
                t.ncol_tiles = tile == G16_T320 ? (int)((g.C + 319) / 320) : (tile == G16_T256P ? (int)((g.C + 255) / 256) : ((tile == G16_T160 || tile == G16_T160N) ? (int)(g.C / 160) : g.ncol_tiles));
                g.worder     = gemm16_worder_rows(t, (unsigned)(rt256 * t.ncol_tiles * mul), ny);
            }
            if (tile == G16_T320) {
                g.ncol_tiles = (int)((g.C + 319) / 320);
#ifdef MI355X_EXPERIMENTS
                if (g_g16_abl == 1) {
                    k_gemm16<256, 320, CONV_, 32, 4, 4, 2, 2><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                    return;
                }
                if (g_g16_abl == 2) {
                    k_gemm16<256, 320, CONV_, 32, 4, 4, 2, 3><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                    return;
                }
                if (g_g16_abl == 3) {  // A tiles fetched for tap 0 only (other taps: the zero page): the DMA volume of a kernel that keeps the input window in LDS
                    k_gemm16<256, 320, CONV_, 32, 4, 4, 2, 4><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                    return;
                }
#endif
                if constexpr (!CONV_) {
                    if (g16_swp_ok(g)) {
                        k_gemm16<256, 320, false, 32, 4, 4, 2, 1, true><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                        return;
                    }
                }
                k_gemm16<256, 320, CONV_, 32, 4, 4, 2, 1><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
            } else if (tile == G16_T256P) {
                g.ncol_tiles = (int)((g.C + 255) / 256);
                if constexpr (!CONV_) {
                    if (g.qt) {  // raw quantised rows, dequantised in the main loop
                        const dim3 grid((unsigned)(rt256 * g.ncol_tiles * mul), ny);
                        if (g.qt == 8)
                            k_gemm16<256, 256, false, 32, 4, 4, 2, 1, false, false, 8><<<grid, 512, 0, s>>>(g);
                        else
                            k_gemm16<256, 256, false, 32, 4, 4, 2, 1, false, false, 4><<<grid, 512, 0, s>>>(g);
                        return;
                    }
                    if (g.C % 256 != 0) g.wblk_lim = (int)(rup64(g.C, 128) / 32);
#ifdef MI355X_EXPERIMENTS
                    if (g_g16_abl == 1) {  // timing ablations of the pipelined 256 x 256 Linear tile (scripts/gemm_ablation.py linear): no MFMAs / no DMA after the fill
                        k_gemm16<256, 256, false, 32, 4, 4, 2, 2><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                        return;
                    }
                    if (g_g16_abl == 2) {
                        k_gemm16<256, 256, false, 32, 4, 4, 2, 3><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                        return;
                    }
                    if (g_g16_abl == 5) {  // ... and no barrier in the steady loop
                        k_gemm16<256, 256, false, 32, 4, 4, 2, 5><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                        return;
                    }
                    if (g_g16_abl == 6) {  // ... and no LDS waits either: the bare issue stream of fragment reads and MFMAs
                        k_gemm16<256, 256, false, 32, 4, 4, 2, 6><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                        return;
                    }
#endif
                    if (g16_swp_ok(g)) {
                        k_gemm16<256, 256, false, 32, 4, 4, 2, 1, true><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                        return;
                    }
                }
                k_gemm16<256, 256, CONV_, 32, 4, 4, 2, 1><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
            } else if (tile == G16_T192P) {
                g.ncol_tiles = (int)((g.C + 191) / 192);
                if (g.C % 192 != 0) g.wblk_lim = (int)(rup64(g.C, 128) / 32);
                if constexpr (!CONV_) {
                    const dim3 grid((unsigned)(rt256 * g.ncol_tiles * mul), ny);
                    if (g.qt == 8)
                        k_gemm16<256, 192, false, 32, 4, 4, 2, 1, false, false, 8><<<grid, 512, 0, s>>>(g);
                    else if (g.qt)
                        k_gemm16<256, 192, false, 32, 4, 4, 2, 1, false, false, 4><<<grid, 512, 0, s>>>(g);
                    else
                        k_gemm16<256, 192, false, 32, 4, 4, 2, 1><<<grid, 512, 0, s>>>(g);
                }
            } else if (tile == G16_T160) {
                g.ncol_tiles = (int)(g.C / 160);
                k_gemm16<256, 160, CONV_, 32, 3, 4, 1><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 256, 0, s>>>(g);
            } else if (tile == G16_T160N) {
                g.ncol_tiles = (int)(g.C / 160);
                if constexpr (!CONV_) {
                    if (g16_swp_ok(g) && g.geglu_inner == 0) {  // 160-column tiles: five column blocks per wave, no GEGLU pairing
                        k_gemm16<256, 160, false, 32, 3, 8, 1, 0, true><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
                        return;
                    }
                }
                k_gemm16<256, 160, CONV_, 32, 3, 8, 1><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
            } else if (tile == G16_T256W) {
                k_gemm16<256, 128, CONV_, 32, 3, 2, 2><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 256, 0, s>>>(g);
            } else {
                k_gemm16<256, 128, CONV_, 32, 3, 4, 2><<<dim3((unsigned)(rt256 * g.ncol_tiles * mul), ny), 512, 0, s>>>(g);
            }
            return;
        }
    }
    const dim3 grid((unsigned)(((rows + 127) / 128) * g.ncol_tiles * mul), ny);
    KScope ks_(s, CONV_ ? KF_CONV_T128 : KF_LINEAR, flops, bytes);
    // (measured and rejected: the hand-pipelined loop of the 256x320 tile instantiated for a 128x128 tile — 2 waves of 64x128, 4 stages, two
    // workgroups per CU — for launches resident in one round: SDXL pair forward 33.6 -> 35.0 ms, SD1.5 step 23.55 -> 23.79 ms; one wave per SIMD
    // hides less than the plain loop's four: profiles/r04s_t128p_rejected_*.txt)
    if (g_g16_variant == 0)
        k_gemm16<128, BN_, CONV_, 64, 2, 2, 2><<<grid, 256, 0, s>>>(g);
    else if (g_g16_variant == 2)
        k_gemm16<128, BN_, CONV_, 64, 3, 2, 2><<<grid, 256, 0, s>>>(g);
    else
        k_gemm16<128, BN_, CONV_, 32, 3, 2, 2><<<grid, 256, 0, s>>>(g);
}

// ---- split-K.  Deep-K contractions over few output tiles (the 8x8 UNet level: 80 tiles on 256 CUs; the 16-row time-embedding
// projections: 10 tiles) leave most of the chip idle.  K is cut into S slices, slice s writes raw partial sums to slab s of a
// workspace in the operand arena (same indexing as dst), and k_splitk_reduce sums the slabs in a fixed order (deterministic,
// unlike float atomics) and applies bias + residual.
static int g_g16_splitk_target = 384;  // option "splitk_target": workgroups a split launch should reach
void gemm16_set_splitk_target(int v) { g_g16_splitk_target = v; }
static int g_g16_splitk_mid = 0;
void gemm16_set_splitk_mid(int v) { g_g16_splitk_mid = v; }
static int g_g16_t320_linear_max_split = 4;  // option "t320_linear_max_split"
void gemm16_set_t320_linear_max_split(int v) { g_g16_t320_linear_max_split = v; }
// K slices for the pipelined 256x320 tile when the output alone does not fill the chip (one workgroup per CU: 256 slots): the 32x32 and
// 16x16 UNet levels give 128 / 64 tiles.  0 = not applicable.  At least 20 stages per slice (4 of them are pipeline fill).
static int g16_t320_split(int64_t rows, int64_t M, int64_t nt, bool conv) {
    if (!g_g16_t320 || g_g16_variant != 3 || M % 320 != 0 || (g_g16_force_tile >= 0 && g_g16_force_tile != G16_T320)) return 0;
    const int64_t c320 = ((rows + 255) / 256) * (M / 320);
    if (c320 >= 192 || c320 < 8) return 0;
    int64_t S = 256 / c320;
    if (S > 16) S = 16;
    // Linears (K <= 5120) with few row tiles: more than four slices means the slab traffic (S f32 outputs written, read again by the reduce pass)
    // outweighs the deeper pipeline — SDXL's 2048-token FF2 / projections: 8 slices -> 128-row tiles with 2 slices, 35.1 -> 33.7 ms of kernels
    // per pair forward (profiles/r04q_sdxl_splitk.txt)
    if (!conv && S > g_g16_t320_linear_max_split) return 0;
    // every slice writes a whole f32 slab and the reduce pass reads them all: worth it only when a slice still carries real work
    // (profiles/r02d_t320_tile_check.txt: K/S = 2880 wins 10-45 %, K/S = 1280 loses 25 %); the 8x8 level (<= 2048 rows) has no good
    // alternative and takes shorter slices
    const int64_t min_stages = rows <= 2048 ? 20 : 48;
    while (S > 1 && nt / S < min_stages) --S;
    return (S >= 2 && c320 * S >= 128) ? (int)S : 0;
}
int gemm16_split_k(int64_t rows, int64_t M, int64_t K, bool conv) {
    if (!g16_bk32()) return 1;
    const int64_t wgs = ((rows + 127) / 128) * ((M + 127) / 128);
    const int64_t nt  = rup64(K, 64) / 32;
    if (const int s320 = g16_t320_split(rows, M, nt, conv)) return s320;
    if (wgs > g_g16_splitk_target / 2) {
        // option "splitk_mid" (experiment, default 0): 193..384 workgroups over 768 resident slots leave most CUs with one or two
        // workgroups (the 16x16 UNet level: 320 tiles, K = 11520..23040 at ~520 TFLOP/s); two K slices double the workgroups in flight
        if (g_g16_splitk_mid && wgs <= g_g16_splitk_target && nt >= 128) return 2;
        return 1;
    }
    int64_t S = g_g16_splitk_target / wgs;
    if (S > 8) S = 8;
    if (S > nt / 8) S = nt / 8;
    return S < 2 ? 1 : (int)S;
}

// ---- split-K policy.  Two mechanisms:
//   * slabs + k_splitk_reduce (above): the 256x320 tile's K slices (g16_t320_split) and any plain-f32-output launch with a tiny grid;
//   * in-launch combine (k_gemm16, BM = 128): launches that leave the chip one 128-row tile per CU or less.  Such a workgroup runs its k-steps
//     as a bare LDS-read -> MFMA chain with nothing to overlap (0.7 us per 128x128x32 step whatever the ring depth, wave count or tile width,
//     profiles/r02q_tile_sweep.txt), so the launch takes nt x 0.7 us; S slices cut the chain to nt / S steps and bring S x the workgroups.
//     The last arriver applies the launch's full epilogue, so every output mode (head-major, f16 image, gate, residual) can split.
static int g_g16_sk_inkernel = 0;  // option "splitk_inkernel": 1 = combine in the launch.  Default off: whole-model A/B on SD1.5 (profiles/r02r_*) 26.36 vs 25.95-26.13 ms of kernels per forward
void gemm16_set_splitk_inkernel(int v) { g_g16_sk_inkernel = v; }
static int g_g16_sk_in_target = 320;  // option "splitk_in_target": workgroups an in-launch split aims for
void gemm16_set_splitk_in_target(int v) { g_g16_sk_in_target = v; }
static int g_g16_bn64_max = 128;  // option "bn64_max_tiles": 64-column tiles for launches of up to this many 128 x 128 tiles
void gemm16_set_bn64_max(int v) { g_g16_bn64_max = v; }
static bool g16_use_bn64(int64_t rows, int64_t M, int mul) {
    const int64_t c128 = ((rows + 127) / 128) * ((M + 127) / 128) * mul;
    return M <= 64 || (M % 64 == 0 && g16_bk32() && (g_g16_force_tile == G16_T128N64 || (g_g16_force_tile < 0 && g_g16_bn64 && c128 <= g_g16_bn64_max)));
}
G16SplitPlan gemm16_split_plan(int64_t rows, int64_t M, int64_t K, bool conv, bool plain_out, bool geglu) {
    G16SplitPlan r{1, false, 0, 0};
    const int64_t nt = rup64(K, 64) / (g16_bk32() ? 32 : 64);
    if (!g16_bk32()) return r;
    if (!conv) {
        int tiles = 0, bn = 0;
        if (const int grid = g16_streamk_grid(rows, M, K, geglu, &tiles, &bn)) {  // S = -grid: stream-K (launch_gemm16_linear / _geglu)
            r.S        = -grid;
            r.inkernel = true;
            r.tiles    = tiles;
            r.ws_bytes = (size_t)grid * 2 * 256 * bn * 4;
            return r;
        }
    }
    if (geglu) return r;  // the GEGLU launch knows no other split
    // splitk_inkernel = 2: only launches whose output mode the slab reduce cannot serve (head-major / f16 / gated epilogues)
    if (g16_t320_split(rows, M, nt, conv) == 0 && (g_g16_sk_inkernel == 1 || (g_g16_sk_inkernel == 2 && !plain_out)) && g_g16_variant == 3 && g_g16_force_tile < 0) {
        const int bn        = (conv ? M <= 64 : g16_use_bn64(rows, M, 1)) ? 64 : 128;
        const int64_t tiles = ((rows + 127) / 128) * ((M + bn - 1) / bn);
        // only grids the 128-row tile would get anyway (g16_pick_tile moves to 256-row tiles from 256 of them on)
        const int64_t c256 = ((rows + 255) / 256) * ((M + 127) / 128);
        if (tiles <= 384 && c256 < 256 && nt >= 16 && tiles < (1 << 20)) {
            int64_t S = (g_g16_sk_in_target + tiles / 2) / tiles;
            if (S > 4) S = 4;
            while (S > 1 && nt / S < 8) --S;
            if (S >= 2) {
                r.S        = (int)S;
                r.inkernel = true;
                r.tiles    = (int)tiles;
                r.ws_bytes = (size_t)tiles * S * 128 * bn * 4;
                return r;
            }
        }
    }
    if (plain_out) {
        r.S = gemm16_split_k(rows, M, K, conv);
        if (r.S > 1) r.ws_bytes = (size_t)r.S * rows * M * 4;
    }
    return r;
}

// dst[i] = sum_s slab_s[i] + bias[(i / inner) % C] + residual[i];  4 elements per thread (n % 4 == 0, and inner % 4 == 0 or inner == 1 with C % 4 == 0)
template <bool V4>
__global__ void k_splitk_reduce(float* __restrict__ dst, const float* __restrict__ ws, int S, int64_t slab, int64_t n, const float* __restrict__ bias,
                                int64_t inner, int C, const float* residual, const float* __restrict__ chan_add, int64_t chan_ld) {
    constexpr int W   = V4 ? 4 : 1;
    const int64_t i   = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
    if (i >= n) return;
    float v[W];
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] = 0.f;
    for (int s = 0; s < S; ++s) {
        if (V4) {
            const float4 t = *(const float4*)(ws + s * slab + i);
            v[0] += t.x; v[W > 1 ? 1 : 0] += t.y; v[W > 2 ? 2 : 0] += t.z; v[W > 3 ? 3 : 0] += t.w;
        } else {
            v[0] += ws[s * slab + i];
        }
    }
    if (bias) {
        if (inner == 1) {
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] += bias[(i + j) % C];
        } else {
            const float b = bias[(i / inner) % C];
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] += b;
        }
    }
    if (chan_add) {  // conv output [OHOW][C][N]: element i belongs to (image, channel) pair i / OHOW
        // (image, channel) pair pc = i / inner = n * C + c sits at chan_add[n * chan_ld + c]
        if (inner == 1) {  // 1x1 feature maps: consecutive elements are consecutive channels (C % 4 == 0 on the vector path: no image boundary inside)
            const int64_t nimg = i / C;
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] += chan_add[nimg * chan_ld + (i - nimg * C) + j];
        } else {
            const int64_t pc = i / inner, nimg = pc / C;
            const float b    = chan_add[nimg * chan_ld + (pc - nimg * C)];
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] += b;
        }
    }
    if (residual) {
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] += residual[i + j];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) dst[i + j] = v[j];
}
// Slab reduce of a split-K conv FUSED with the statistics of the GroupNorm that reads the result next: one workgroup per (image, group) — the
// group's cpg * hw outputs are one contiguous NCHW run — sums the S slabs in slice order, adds bias / embedding / residual exactly as
// k_splitk_reduce does (same operation order: identical values), stores them, and keeps them in registers for the two-pass mean / variance of
// k_gn_stats_reg (same per-thread order, same block reductions), then writes the per-(image, channel) affine.  Saves the GroupNorm
// statistics pass (one more read of the tensor) and its launch for every split conv that feeds a GroupNorm.
template <int NT, int NV>
__global__ __launch_bounds__(NT) void k_splitk_reduce_gn(float* __restrict__ dst, const float* __restrict__ ws, int S, int64_t slab, int64_t hw, int C, int groups, int cpg,
                                                         const float* __restrict__ bias, const float* residual, const float* __restrict__ chan_add, int64_t chan_ld, float eps,
                                                         const float* __restrict__ gw, const float* __restrict__ gb, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float scratch[2 * (NT / 64)];
    const int gidx = blockIdx.x % groups, n = blockIdx.x / groups;
    const int c0   = gidx * cpg;
    const int64_t cnt = (int64_t)cpg * hw, base = ((int64_t)n * C + c0) * hw, n4 = cnt / 4;
    float4 v[NV];
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t i4 = threadIdx.x + (int64_t)NT * j;
        v[j]             = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < n4) {
            const int64_t e = i4 * 4, i = base + e;
            float4 a        = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int sl = 0; sl < S; ++sl) {
                const float4 t = *(const float4*)(ws + sl * slab + i);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            const int c = c0 + (int)(e / hw);  // hw % 4 == 0: the four elements share their channel
            if (bias) {
                const float b = bias[c];
                a.x += b; a.y += b; a.z += b; a.w += b;
            }
            if (chan_add) {
                const float b = chan_add[(int64_t)n * chan_ld + c];
                a.x += b; a.y += b; a.z += b; a.w += b;
            }
            if (residual) {
                const float4 r = *(const float4*)(residual + i);
                a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
            }
            *(float4*)(dst + i) = a;
            v[j]                = a;
        }
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    block_sum2<NT / 64>(s, dummy, scratch);
    const float mean = s / (float)cnt;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (threadIdx.x + (int64_t)NT * j < n4) {
            const float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    dummy = 0.f;
    block_sum2<NT / 64>(q, dummy, scratch);
    const float rstd = rsqrtf(q / (float)cnt + eps);
    for (int c = c0 + threadIdx.x; c < c0 + cpg; c += NT) {
        const float sc            = (gw ? gw[c] : 1.f) * rstd;
        scale[(int64_t)n * C + c] = sc;
        shift[(int64_t)n * C + c] = (gb ? gb[c] : 0.f) - mean * sc;
    }
}
// true when the fused pass serves this launch: whole groups on the vector path that fit the register-resident variants, and at least one workgroup
// per CU — the pass has ONE workgroup per (image, group), and with fewer it loses more on the slab sums than the statistics pass costs (SDXL's
// batch-1 cfg pair, 64 workgroups: reduce 2.79 -> 3.48 ms against 0.27 ms of statistics saved, profiles/r04G_*)
bool splitk_reduce_gn_supported(int64_t hw, int64_t C, int64_t N, int groups) {
    if (groups <= 0 || C % groups != 0 || hw % 4 != 0 || N * groups < 256) return false;
    return (C / groups) * hw <= 4 * 1024 * 16;
}
static void launch_splitk_reduce_gn(hipStream_t s, float* dst, const float* ws, int S, int64_t hw, int64_t C, int64_t N, const Epilogue& e) {
    const int groups = e.gn_groups, cpg = (int)(C / groups);
    const int64_t cnt = (int64_t)cpg * hw, n = hw * C * N;
    const int64_t chan_ld = e.chan_ld > 0 ? e.chan_ld : C;
    KScope ks_(s, KF_SPLITK, 0.0, (double)n * 4.0 * (S + 1 + (e.residual ? 1 : 0)));
    const unsigned grid = (unsigned)(N * groups);
#define SKGN(NT_, NV_) k_splitk_reduce_gn<NT_, NV_><<<grid, NT_, 0, s>>>(dst, ws, S, n, hw, (int)C, groups, cpg, e.bias, e.residual, e.chan_add, chan_ld, e.gn_eps, e.gn_w, e.gn_b, e.gn_scale, e.gn_shift)
    if (cnt <= 4 * 256 * 4)  // the same variant choice as launch_gn_stats: same summation order, same statistics
        SKGN(256, 4);
    else if (cnt <= 4 * 1024 * 4)
        SKGN(1024, 4);
    else
        SKGN(1024, 16);
#undef SKGN
}

static void launch_splitk_reduce(hipStream_t s, float* dst, const float* ws, int S, int64_t n, const float* bias, int64_t inner, int64_t C, const float* residual,
                                 const float* chan_add = nullptr, int64_t chan_ld = 0) {
    if (chan_ld <= 0) chan_ld = C;
    KScope ks_(s, KF_SPLITK, 0.0, (double)n * 4.0 * (S + 1));
    const bool v4 = n % 4 == 0 && (inner % 4 == 0 || (inner == 1 && C % 4 == 0)) && (((uintptr_t)dst | (uintptr_t)ws | (uintptr_t)residual) & 15) == 0;
    if (v4)
        k_splitk_reduce<true><<<(unsigned)((n / 4 + 255) / 256), 256, 0, s>>>(dst, ws, S, n, n, bias, inner, (int)C, residual, chan_add, chan_ld);
    else
        k_splitk_reduce<false><<<(unsigned)((n + 255) / 256), 256, 0, s>>>(dst, ws, S, n, n, bias, inner, (int)C, residual, chan_add, chan_ld);
}

// Slab reduce of a split-K Linear FUSED with the LayerNorm that reads the result next (Epilogue::ln_*): one wave per row (NV float4 per lane,
// M <= 256 * NV), S <= 4 slabs summed in slice order, bias and residual added in the order of k_splitk_reduce (identical values), the f32 row
// stored, then normalised from registers exactly as k_layer_norm_f16_reg does (same per-lane order, same wave reductions: identical image).
// Saves the LayerNorm launch and its read of the tensor: at SDXL's 32x32 level (2048 rows) both passes are ~11 us launches bound by their latency
// chain, three pairs per transformer block.  (First version, round 4: four rows per wave on 16-lane groups — 128 workgroups for 2048 rows and
// 100 dependent loads per lane: 45 us per launch instead of 11.7 + 10.6, profiles/r05b_*; the standalone LayerNorm in that layout was also
// slower, 2.28 -> 4.55 ms per SDXL forward, and was removed.)
template <int NV, int S>
__global__ __launch_bounds__(256) void k_splitk_reduce_ln(float* __restrict__ dst, const float* __restrict__ ws, int64_t slab, int64_t nrows, int M, int Kp,
                                                          const float* __restrict__ bias, const float* residual, _Float16* __restrict__ dst16, float eps,
                                                          const float* __restrict__ w, const float* __restrict__ b) {
    const int lane    = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int n4 = M / 4;
    float4 v[NV], t[S][NV], r[NV];
    // every load of the row is issued before the first add
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float4* p = (const float4*)(ws + s * slab + row * M);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            t[s][j]     = i < n4 ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        r[j]        = (residual && i < n4) ? ((const float4*)(residual + row * M))[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        v[j]        = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            v[j].x += t[s][j].x; v[j].y += t[s][j].y; v[j].z += t[s][j].z; v[j].w += t[s][j].w;
        }
        if (i >= n4) continue;
        if (bias) { const float4 bb = ((const float4*)bias)[i]; v[j].x += bb.x; v[j].y += bb.y; v[j].z += bb.z; v[j].w += bb.w; }
        if (residual) { v[j].x += r[j].x; v[j].y += r[j].y; v[j].z += r[j].z; v[j].w += r[j].w; }
        ((float4*)(dst + row * M))[i] = v[j];
    }
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);  // slots beyond n4 hold zeros
    const float mean = wave_sum(sm) / (float)M;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (lane + 64 * j < n4) {
            const float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)M + eps);
    _Float16* yr = dst16 + row * Kp;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i >= Kp / 4) continue;
        half4_t h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (i < n4) {
            float4 x = v[j];
            x.x = (x.x - mean) * rstd; x.y = (x.y - mean) * rstd; x.z = (x.z - mean) * rstd; x.w = (x.w - mean) * rstd;
            if (w) { const float4 ww = ((const float4*)w)[i]; x.x *= ww.x; x.y *= ww.y; x.z *= ww.z; x.w *= ww.w; }
            if (b) { const float4 bb = ((const float4*)b)[i]; x.x += bb.x; x.y += bb.y; x.z += bb.z; x.w += bb.w; }
            h[0] = (_Float16)x.x; h[1] = (_Float16)x.y; h[2] = (_Float16)x.z; h[3] = (_Float16)x.w;
        }
        *(half4_t*)(yr + i * 4) = h;
    }
}
bool splitk_reduce_ln_supported(int64_t rows, int64_t M) { return M % 4 == 0 && rup64(M, 64) <= 1280 && rows >= 64; }
static void launch_splitk_reduce_ln(hipStream_t s, float* dst, const float* ws, int S, int64_t rows, int64_t M, const Epilogue& e) {
    if (!splitk_reduce_ln_supported(rows, M) || S < 2 || S > 4 ||
        ((((uintptr_t)dst | (uintptr_t)ws | (uintptr_t)e.residual | (uintptr_t)e.bias | (uintptr_t)e.ln_w | (uintptr_t)e.ln_b)) & 15) != 0) {
        // the planner registered this LayerNorm as done (plan_linear look-ahead checks the same conditions with the same addresses)
        fprintf(stderr, "ggml-mi355x: split-K reduce asked for a LayerNorm image on a shape / slice count / alignment it does not serve\n");
        abort();
    }
    const int Kp = (int)rup64(M, 64);
    // algorithmic bytes: the slabs + residual in, the f32 tensor and the f16 image out
    KScope ks_(s, KF_SPLITK, 0.0, (double)rows * M * 4.0 * (S + 1 + (e.residual ? 1 : 0)) + (double)rows * Kp * 2.0);
    const unsigned grid = (unsigned)((rows + 3) / 4);
#define SKLN(NV_, S_) k_splitk_reduce_ln<NV_, S_><<<grid, 256, 0, s>>>(dst, ws, rows * M, rows, (int)M, Kp, e.bias, e.residual, (_Float16*)e.ln_dst16, e.ln_eps, e.ln_w, e.ln_b)
#define SKLN_S(NV_)             \
    do {                        \
        if (S == 2)             \
            SKLN(NV_, 2);       \
        else if (S == 3)        \
            SKLN(NV_, 3);       \
        else                    \
            SKLN(NV_, 4);       \
    } while (0)
    if (Kp <= 256 * 2)
        SKLN_S(2);
    else if (Kp <= 256 * 3)
        SKLN_S(3);
    else
        SKLN_S(5);
#undef SKLN_S
#undef SKLN
}

void splitk_reduce_rows(hipStream_t s, float* dst, const float* ws, int S, int64_t n, const float* bias, int64_t C, const float* residual) {
    launch_splitk_reduce(s, dst, ws, S, n, bias, 1, C, residual);
}

// a 256-byte zero page per device for the padding taps: ONE process-wide allocation per device, made by gemm16_init() (planner_create) — never inside a
// launch function (round-5 advice: a thread-local page allocated lazily ran hipMalloc / hipMemset inside a stream capture when the capturing thread was
// not the one that had run the plan eagerly, which fails the capture)
static std::atomic<const _Float16*> g_zero_page[64];
static std::mutex g_zero_mu;
static const _Float16* zero_page() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const _Float16* z = g_zero_page[dev & 63].load(std::memory_order_acquire);
    if (z) return z;
    std::lock_guard<std::mutex> lk(g_zero_mu);
    z = g_zero_page[dev & 63].load(std::memory_order_relaxed);
    if (!z) {
        void* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) {
            fprintf(stderr, "ggml-mi355x: zero page allocation failed on device %d (gemm16_init must run before the first launch, outside any stream capture)\n", dev);
            abort();
        }
        (void)hipDeviceSynchronize();
        z = (const _Float16*)p;
        g_zero_page[dev & 63].store(z, std::memory_order_release);
    }
    return z;
}

void gemm16_init() { (void)zero_page(); }
const _Float16* gemm16_zero_page() { return zero_page(); }
// The planner sets e.gn_* only after checking shape AND alignment with the launch's own addresses (plan_conv_chain::gn_register) and then drops
// the GroupNorm's statistics pass; a launch that cannot honour that would leave the scale / shift tables unwritten — stop instead of going on.
[[noreturn]] static void g16_gn_contract_violation() {
    fprintf(stderr, "ggml-mi355x: split-K reduce asked for GroupNorm statistics on a shape / alignment it does not serve\n");
    abort();
}
void launch_splitk_reduce_conv(hipStream_t s, float* dst, const float* ws, int S, int64_t n, const float* bias, int64_t inner, int64_t C, const float* residual,
                               const float* chan_add, int64_t chan_ld) {
    launch_splitk_reduce(s, dst, ws, S, n, bias, inner, C, residual, chan_add, chan_ld);
}
// the same with the statistics of the GroupNorm that reads the output next (e.gn_*); falls back to the plain pass when the shape is not served
void launch_splitk_reduce_conv_gn(hipStream_t s, float* dst, const float* ws, int S, int64_t hw, int64_t C, int64_t N, const Epilogue& e) {
    const bool gn_ok = (((uintptr_t)dst | (uintptr_t)ws | (uintptr_t)e.residual) & 15) == 0 && splitk_reduce_gn_supported(hw, C, N, e.gn_groups);
    if (e.gn_scale && !gn_ok) g16_gn_contract_violation();
    if (e.gn_scale)
        launch_splitk_reduce_gn(s, dst, ws, S, hw, C, N, e);
    else
        launch_splitk_reduce(s, dst, ws, S, hw * C * N, e.bias, hw, C, e.residual, e.chan_add, e.chan_ld);
}

// GGML_MI355X_TRACE=1: one stderr line per launch (shape), in launch order — joined with a rocprofv3 kernel trace by scripts/shape_stats.py
static bool g16_trace() {
    static const bool on = getenv("GGML_MI355X_TRACE") != nullptr;
    return on;
}
static void g16_check_epi(const Epilogue& e) {
    if (e.act >= 0) {
        fprintf(stderr, "ggml-mi355x: gemm16 kernels have no fused activation (act=%d requested)\n", e.act);
        abort();
    }
}

void launch_gemm16_linear(hipStream_t s, float* dst, void* dst16, int64_t ldd16, const void* a16, int64_t lda, const void* wswz, int64_t rows, int64_t K,
                          int64_t M, int64_t ldd, const Epilogue& e, int hm_d, int hm_H, int hm_L, float* splitk_ws, int* splitk_cnt, int splitk_S) {
    G16Args g{};
    g.A     = (const _Float16*)a16;
    g.lda   = lda;
    g.W     = (const half8_t*)wswz;
    const int64_t Kp = rup64(K, 64);
    g.kfr   = Kp / 16;
    g.dst   = dst;
    g.dst16 = (_Float16*)dst16;
    g.ldd   = ldd;
    g.ldd16 = ldd16;
    if (hm_d > 0 && (hm_L < 32 || e.residual || (dst != nullptr) == (dst16 != nullptr) || rows >= (1ll << 31))) {
        fprintf(stderr, "ggml-mi355x: head-major gemm16 store needs L >= 32, exactly one output and no residual\n");
        abort();
    }
    g.hm_d  = hm_d;
    g.hm_H  = hm_H;
    g.hm_L  = hm_L;
    g.R     = rows;
    g.C     = M;
    g.nt    = (int)(Kp / (g16_bk32() ? 32 : 64));
    g16_check_epi(e);
#ifdef MI355X_EXPERIMENTS
    g.abl = g_g16_abl;
#endif
    g.ep    = {e.bias, e.residual, e.scale, e.gate, e.gate_L, e.gelu};
    if (e.a_run_L > 0 && e.a_run_S != e.a_run_L) {
        if (rows >= (1ll << 31) || e.a_run_S >= (1ll << 31) || rows % e.a_run_L != 0) {
            fprintf(stderr, "ggml-mi355x: invalid operand-run request (rows %lld, run %lld, stride %lld)\n", (long long)rows, (long long)e.a_run_L, (long long)e.a_run_S);
            abort();
        }
        g.a_runL = (int)e.a_run_L;
        g.a_runS = (int)e.a_run_S;
    }
    if (e.qtype) {
        g.qt         = e.qtype == 8 ? 8 : 4;
        g.qrow_bytes = e.qrow_bytes;
        if ((e.qtype != 8 && e.qtype != 2) || e.qrow_bytes % 4 != 0 || ((uintptr_t)wswz & 15) != 0 || K % 64 != 0) {
            fprintf(stderr, "ggml-mi355x: invalid in-loop dequantisation request (type %d, row bytes %lld)\n", e.qtype, (long long)e.qrow_bytes);
            abort();
        }
    }
    if (e.split_col > 0) {
        if (!gemm16_split_col_supported(rows, M, K) || !dst || dst16 || hm_d > 0 || e.gate || e.gelu || e.residual || !e.split_dst16 || e.split_col % 256 != 0 || e.split_col >= M || M % 160 == 0 || (splitk_ws && splitk_S > 1)) {
            fprintf(stderr, "ggml-mi355x: invalid column-range epilogue request (split_col)\n");
            abort();
        }
        g.split_col   = (int)e.split_col;
        g.split_dst16 = (_Float16*)e.split_dst16 - e.split_col;  // indexed by the global column
        g.split_ldd16 = e.split_ldd16;
    }
    if ((e.gate && (!e.residual || e.gate_L < 32 || !dst || dst16 || hm_d > 0 || rows >= (1ll << 31))) || (e.gelu && (dst || !dst16 || e.residual || hm_d > 0))) {
        fprintf(stderr, "ggml-mi355x: invalid gated / gelu gemm16 epilogue request\n");
        abort();
    }
    const bool streamk = splitk_ws && splitk_cnt != nullptr && splitk_S < 0;
    if (streamk) {  // gemm16_split_plan's S = -grid
        int tiles = 0, dp = 0;
        if (g16_streamk_grid(rows, M, K, false, &tiles, nullptr, &dp) != -splitk_S) {
            fprintf(stderr, "ggml-mi355x: stream-K plan and launch disagree (options changed between plan and launch?)\n");
            abort();
        }
        g.sk_grid  = -splitk_S;
        g.sk_tiles = tiles;
        g.sk_dp    = dp;
        g.sk_slab  = splitk_ws;
        g.sk_cnt   = splitk_cnt;
    }
    const bool inker = !streamk && splitk_ws && splitk_cnt != nullptr && splitk_S > 1;
    const int S      = (streamk || e.split_col > 0) ? 1 : inker ? splitk_S : ((splitk_ws && dst && !dst16 && hm_d == 0 && ldd == M && !e.gate) ? gemm16_split_k(rows, M, K, false) : 1);
    if (S > 1) {
        g.split_k  = S;
        g.nt_slice = (g.nt + S - 1) / S;
        if (inker) {  // the last arriver of every tile applies the epilogue itself
            g.sk_slab = splitk_ws;
            g.sk_cnt  = splitk_cnt;
        } else {
            g.slab = rows * M;
            g.dst  = splitk_ws;
            g.ep   = G16Epi{nullptr, nullptr, e.scale};
        }
    }
    // 64-column tiles: narrow outputs, and small grids (<= 128 tiles of 128x128 on 256 CUs: the cross-attention K/V projections of the 77-token
    // context, the time-embedding Linears) where twice the workgroups matter more than the tile's arithmetic intensity (r02q: 1232x768->768 36 -> 21 us)
    const bool bn64 = g16_use_bn64(rows, M, 1);
    // algorithmic bytes: A image + weight image once, output once (f32 or f16), residual once
    const double lin_bytes = (double)rows * rup64(K, 64) * 2.0 + (double)rup64(K, 64) * rup64(M, 128) * 2.0 + (double)rows * M * (dst16 ? 2.0 : 4.0) + (e.residual ? (double)rows * M * 4.0 : 0.0);
    if (g16_trace()) fprintf(stderr, "G16 linear rows=%lld K=%lld M=%lld res=%d hm=%d f16out=%d\n", (long long)rows, (long long)K, (long long)M, e.residual ? 1 : 0, hm_d, dst16 ? 1 : 0);
    if (bn64) {
        g.ncol_tiles = (int)((M + 63) / 64);
        g16_launch<64, false>(s, g, rows, 2.0 * rows * K * M, lin_bytes);
    } else {
        g.ncol_tiles = (int)((M + 127) / 128);
        g16_launch<128, false>(s, g, rows, 2.0 * rows * K * M, lin_bytes);
    }
    if (e.ln_dst16 && !(S > 1 && !inker)) {
        fprintf(stderr, "ggml-mi355x: a LayerNorm image was asked of a Linear that does not run the split-K reduce pass\n");
        abort();
    }
    if (S > 1 && !inker) {
        if (e.ln_dst16)
            launch_splitk_reduce_ln(s, dst, splitk_ws, S, rows, M, e);
        else
            launch_splitk_reduce(s, dst, splitk_ws, S, rows * M, e.bias, 1, M, e.residual);
    }
}

// n (2..16) sibling Linears over the same f16 operand image in ONE launch (q / k / v projections of a self-attention, k / v of a cross-attention:
// block.hpp CrossAttention): same rows, K, M, head-major parameters and scale; per weight its image, f32 and / or f16 destination and bias.  The
// operand rows are fetched once per row tile (the column tiles of all weights run back to back on one XCD) and two or three ~25 us launch chains
// become one.
void launch_gemm16_linear_multi(hipStream_t s, int n, float* const* dst, void* const* dst16, const void* a16, int64_t lda, const void* const* wswz, int64_t rows,
                               int64_t K, int64_t M, const float* const* bias, float scale, int hm_d, int hm_H, int hm_L) {
    if (n < 2 || n > 16) {
        fprintf(stderr, "ggml-mi355x: launch_gemm16_linear_multi takes 2..16 weights\n");
        abort();
    }
    G16Args g{};
    g.A     = (const _Float16*)a16;
    g.lda   = lda;
    const int64_t Kp = rup64(K, 64);
    g.kfr   = Kp / 16;
    g.ldd   = M;
    g.ldd16 = 0;
    g.hm_d  = hm_d;
    g.hm_H  = hm_H;
    g.hm_L  = hm_L;
    g.R     = rows;
    g.C     = M;
    g.nt    = (int)(Kp / (g16_bk32() ? 32 : 64));
    g.ep    = G16Epi{bias[0], nullptr, scale};
    g.multi = n;
    for (int i = 0; i < n; ++i) {
        g.Wm[i]     = (const half8_t*)wswz[i];
        g.dstm[i]   = dst[i];
        g.dst16m[i] = (_Float16*)dst16[i];
        g.biasm[i]  = bias[i];
    }
    g.W     = g.Wm[0];
    g.dst   = g.dstm[0];
    g.dst16 = g.dst16m[0];
    double out_b = 0.0;
    for (int i = 0; i < n; ++i) out_b += (double)rows * M * (dst16[i] ? 2.0 : 4.0);
    const double bytes = (double)rows * Kp * 2.0 + (double)n * Kp * rup64(M, 128) * 2.0 + out_b;
    if (g16_trace()) fprintf(stderr, "G16 linear x%d rows=%lld K=%lld M=%lld hm=%d\n", n, (long long)rows, (long long)K, (long long)M, hm_d);
    if (g16_use_bn64(rows, M, n)) {
        g.ncol_tiles = (int)((M + 63) / 64);
        g16_launch<64, false>(s, g, rows, 2.0 * n * rows * K * M, bytes);
    } else {
        g.ncol_tiles = (int)((M + 127) / 128);
        g16_launch<128, false>(s, g, rows, 2.0 * n * rows * K * M, bytes);
    }
}

void launch_gemm16_linear_geglu(hipStream_t s, void* dst16, const void* a16, int64_t lda, const void* wswz_geglu, int64_t rows, int64_t K, int64_t M,
                                const float* bias, float* splitk_ws, int* splitk_cnt, int splitk_S, int geglu_mode) {
    G16Args g{};
    g.A           = (const _Float16*)a16;
    g.lda         = lda;
    g.W           = (const half8_t*)wswz_geglu;
    const int64_t Kp = rup64(K, 64);
    g.kfr         = Kp / 16;
    g.dst16       = (_Float16*)dst16;
    g.geglu_inner = (int)(M / 2);
    g.geglu16     = geglu_mode == 2 ? 1 : 0;  // the layout the planner built the weight image in (gemm16_geglu_mode)
    g.ldd16       = M / 2;
    g.R           = rows;
    g.C           = M;
    g.nt          = (int)(Kp / (g16_bk32() ? 32 : 64));
    g.ep          = G16Epi{bias, nullptr, 1.f};
    g.ncol_tiles  = (int)((M + 127) / 128);
#ifdef MI355X_EXPERIMENTS
    g.abl = g_g16_abl;
#endif
    if (splitk_ws && splitk_cnt && splitk_S < 0) {  // stream-K (gemm16_split_plan(..., geglu = true))
        int tiles = 0, dp = 0;
        if (g16_streamk_grid(rows, M, K, true, &tiles, nullptr, &dp) != -splitk_S) {
            fprintf(stderr, "ggml-mi355x: stream-K plan and launch disagree (options changed between plan and launch?)\n");
            abort();
        }
        g.sk_grid  = -splitk_S;
        g.sk_tiles = tiles;
        g.sk_dp    = dp;
        g.sk_slab  = splitk_ws;
        g.sk_cnt   = splitk_cnt;
    }
    if (g16_trace()) fprintf(stderr, "G16 linear rows=%lld K=%lld M=%lld res=0 hm=0 f16out=1 geglu=1\n", (long long)rows, (long long)K, (long long)M);
    g16_launch<128, false>(s, g, rows, 2.0 * rows * K * M, (double)rows * rup64(K, 64) * 2.0 + (double)rup64(K, 64) * rup64(M, 128) * 2.0 + (double)rows * (M / 2) * 2.0);
}

void launch_gemm16_conv(hipStream_t s, float* dst, const void* x16_nhwc, const void* wswz, int64_t W, int64_t H, int64_t IC, int64_t N, int64_t OC, int ksize,
                        int stride, int pad, bool upscale2x, const Epilogue& e, float* splitk_ws, int* splitk_cnt, int splitk_S) {
    G16Args g{};
    g.A   = (const _Float16*)x16_nhwc;
    g.W   = (const half8_t*)wswz;
    g.ICp = (int)rup64(IC, 64);
    g.kfr = (int64_t)g.ICp * ksize * ksize / 16;
    g.dst = dst;
    g.H   = (int)H;
    g.Wd  = (int)W;
    const int CW = upscale2x ? (int)W * 2 : (int)W, CH = upscale2x ? (int)H * 2 : (int)H;
    g.OW   = (CW + 2 * pad - ksize) / stride + 1;
    g.OH   = (CH + 2 * pad - ksize) / stride + 1;
    g.OHOW = (int64_t)g.OW * g.OH;
    g.S    = stride;
    g.pad  = pad;
    g.UPS  = upscale2x ? 1 : 0;
    g.KS   = ksize;
    g.icb_per_tap = g.ICp / 64;
    g.tap_major   = g_g16_tap_major;
    g.nt   = ksize * ksize * g.icb_per_tap * (g16_bk32() ? 2 : 1);
    g.R    = g.OHOW * N;
    g.C    = OC;
    g.zero = zero_page();
    g16_check_epi(e);
    g.ep   = G16Epi{e.bias, e.residual, e.scale};
    g.ep.chan_add = e.chan_add;
    g.ep.chan_ld  = (int)e.chan_ld;
    const bool inker = splitk_ws && splitk_cnt != nullptr && splitk_S > 1;
    const int S      = inker ? splitk_S : (splitk_ws ? gemm16_split_k(g.R, OC, (int64_t)g.ICp * ksize * ksize, true) : 1);
    if (S > 1) {
        g.split_k  = S;
        g.nt_slice = (g.nt + S - 1) / S;
        if (inker) {
            g.sk_slab = splitk_ws;
            g.sk_cnt  = splitk_cnt;
        } else {
            g.slab = g.R * OC;
            g.dst  = splitk_ws;
            g.ep   = G16Epi{nullptr, nullptr, e.scale};
        }
    }
    const bool bn64 = OC <= 64;
    // algorithmic bytes: the NHWC f16 input image once, the weight image once, the f32 output once (+ residual once)
    const double conv_bytes = (double)N * H * W * g.ICp * 2.0 + (double)g.ICp * ksize * ksize * rup64(OC, 128) * 2.0 + (double)g.R * OC * 4.0 * (e.residual ? 2.0 : 1.0);
    if (g16_trace())
        fprintf(stderr, "G16 conv rows=%lld K=%lld M=%lld res=%d ks=%d s=%d ups=%d hw=%lldx%lld ic=%lld\n", (long long)g.R, (long long)g.ICp * ksize * ksize, (long long)OC,
                e.residual ? 1 : 0, ksize, stride, g.UPS, (long long)W, (long long)H, (long long)IC);
    if (bn64) {
        g.ncol_tiles = (int)((OC + 63) / 64);
        g16_launch<64, true>(s, g, g.R, 2.0 * g.R * IC * ksize * ksize * OC, conv_bytes);
    } else {
        g.ncol_tiles = (int)((OC + 127) / 128);
        g16_launch<128, true>(s, g, g.R, 2.0 * g.R * IC * ksize * ksize * OC, conv_bytes);
    }
    if (S > 1 && !inker) {
        const bool gn_ok = (((uintptr_t)dst | (uintptr_t)splitk_ws | (uintptr_t)e.residual) & 15) == 0 && splitk_reduce_gn_supported(g.OHOW, OC, N, e.gn_groups);
        if (e.gn_scale && !gn_ok) g16_gn_contract_violation();
        if (e.gn_scale)
            launch_splitk_reduce_gn(s, dst, splitk_ws, S, g.OHOW, OC, N, e);
        else
            launch_splitk_reduce(s, dst, splitk_ws, S, g.R * OC, e.bias, g.OHOW, OC, e.residual, e.chan_add, e.chan_ld);
    }
}

// =====================================================================================================
// producers of the f16 operand images
// =====================================================================================================
// f32 rows [R][K] (row stride xs floats) -> f16 [R][Kp], zero padded
// rows r = (n, l): n = r / L images with element stride bs, l = r % L rows with stride xs (a sliced token range keeps its parent's batch stride)
__global__ void k_pack_rows_f16(_Float16* __restrict__ dst, const float* __restrict__ xin, int64_t R, int K, int Kp, int64_t xs, int64_t L, int64_t bs) {
    const int64_t n8 = R * (Kp / 8);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (Kp / 8);
        const int k0    = (int)(i - r * (Kp / 8)) * 8;
        const float* x  = xin + (r / L) * bs + (r % L) * xs - r * xs;  // so that x + r * xs addresses row r
        half8_t h;
        if (k0 + 8 <= K) {
            const float4 a = *(const float4*)(x + r * xs + k0), b = *(const float4*)(x + r * xs + k0 + 4);
            h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
            h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (_Float16)(k0 + j < K ? x[r * xs + k0 + j] : 0.f);
        }
        *(half8_t*)(dst + r * Kp + k0) = h;
    }
}
void launch_pack_rows_f16(hipStream_t s, void* dst, const float* x, int64_t R, int64_t K, int64_t xs, int64_t L, int64_t bs) {
    KScope ks_(s, KF_PACK_F16, 0.0, (double)R * K * 4.0 + (double)R * rup64(K, 64) * 2.0);
    const int Kp = (int)rup64(K, 64);
    const int64_t n8 = R * (Kp / 8);
    int64_t blocks   = (n8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (L <= 0) {  // one run of rows
        L  = R;
        bs = 0;
    }
    k_pack_rows_f16<<<(unsigned)blocks, 256, 0, s>>>((_Float16*)dst, x, R, (int)K, Kp, xs, L, bs);
}

// f32 rows [R][K] (row stride xs floats) -> columns [0, K) of f16 rows with stride ld halfs (dst already points at the first column): one part of an
// operand image that several producers fill side by side (concat along the feature dimension feeding a Linear).  GELU = 1: tanh-GELU first
// (FLUX single block, flux.hpp:594-700: gelu(mlp) next to the attention output in linear2's operand).  K % 8 == 0, 16-byte aligned rows.
template <int GELU>
__global__ void k_pack_cols_f16(_Float16* __restrict__ dst, int64_t ld, const float* __restrict__ x, int64_t R, int K, int64_t xs) {
    const int K8     = K / 8;
    const int64_t n8 = R * K8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / K8;
        const int k0    = (int)(i - r * K8) * 8;
        float4 a = *(const float4*)(x + r * xs + k0), b = *(const float4*)(x + r * xs + k0 + 4);
        if (GELU) {
            a.x = act_apply<UN_GELU>(a.x); a.y = act_apply<UN_GELU>(a.y); a.z = act_apply<UN_GELU>(a.z); a.w = act_apply<UN_GELU>(a.w);
            b.x = act_apply<UN_GELU>(b.x); b.y = act_apply<UN_GELU>(b.y); b.z = act_apply<UN_GELU>(b.z); b.w = act_apply<UN_GELU>(b.w);
        }
        half8_t h;
        h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
        h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
        *(half8_t*)(dst + r * ld + k0) = h;
    }
}
void launch_pack_cols_f16(hipStream_t s, void* dst, int64_t ld, const float* x, int64_t R, int64_t K, int64_t xs, bool gelu) {
    KScope ks_(s, KF_PACK_F16, 0.0, (double)R * K * 6.0);
    const int64_t n8 = R * (K / 8);
    int64_t blocks   = (n8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) return;
    if (gelu)
        k_pack_cols_f16<1><<<(unsigned)blocks, 256, 0, s>>>((_Float16*)dst, ld, x, R, (int)K, xs);
    else
        k_pack_cols_f16<0><<<(unsigned)blocks, 256, 0, s>>>((_Float16*)dst, ld, x, R, (int)K, xs);
}

// LayerNorm / RMSNorm (+affine) writing the f16 operand image: one wave per row
// mod_L > 0: adaLN modulate (mmdit.hpp:368-380) — w and b are per-image [images][ne0] tables and the affine is norm * (1 + w) + b
__global__ __launch_bounds__(256) void k_layer_norm_f16(_Float16* __restrict__ dst, const float* __restrict__ x, int ne0, int Kp, int64_t nrows, int64_t xs,
                                                        float eps, const float* __restrict__ w, const float* __restrict__ b, int rms, int64_t mod_L) {
    const int lane    = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float wadd = mod_L > 0 ? 1.f : 0.f;
    if (mod_L > 0) {
        w += (row / mod_L) * ne0;
        b += (row / mod_L) * ne0;
    }
    const float* xr = x + row * xs;
    _Float16* yr    = dst + row * Kp;
    const int n4    = ne0 / 4;  // callers guarantee ne0 % 4 == 0 and 16-byte alignment
    float mean = 0.f;
    if (!rms) {
        float s = 0.f;
        for (int i = lane; i < n4; i += 64) {
            const float4 v = ((const float4*)xr)[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
        mean = wave_sum(s) / (float)ne0;
    }
    float q = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = ((const float4*)xr)[i];
        const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += (a * a + bb * bb) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)ne0 + eps);
    for (int i = lane; i < Kp / 4; i += 64) {
        half4_t h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (i < n4) {
            float4 v = ((const float4*)xr)[i];
            v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd; v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
            if (w) { const float4 ww = ((const float4*)w)[i]; v.x *= ww.x + wadd; v.y *= ww.y + wadd; v.z *= ww.z + wadd; v.w *= ww.w + wadd; }
            if (b) { const float4 bb = ((const float4*)b)[i]; v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
            h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        }
        *(half4_t*)(yr + i * 4) = h;
    }
}
// register-resident variant: a wave keeps its row in NV float4 registers per lane (ne0 <= 256 * NV), so the row is read from memory ONCE
// (the generic kernel above walks it three times: mean, variance, write); same summation order per lane, same wave reductions
template <int NV>
__global__ __launch_bounds__(256) void k_layer_norm_f16_reg(_Float16* __restrict__ dst, const float* __restrict__ x, int ne0, int Kp, int64_t nrows, int64_t xs,
                                                            float eps, const float* __restrict__ w, const float* __restrict__ b, int rms, int64_t mod_L) {
    const int lane    = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float wadd = mod_L > 0 ? 1.f : 0.f;
    if (mod_L > 0) {
        w += (row / mod_L) * ne0;
        b += (row / mod_L) * ne0;
    }
    const float4* xr = (const float4*)(x + row * xs);
    _Float16* yr     = dst + row * Kp;
    const int n4     = ne0 / 4;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        v[j]        = i < n4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mean = 0.f;
    if (!rms) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        mean = wave_sum(s) / (float)ne0;
    }
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (lane + 64 * j < n4) {
            const float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)ne0 + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i >= Kp / 4) continue;
        half4_t h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (i < n4) {
            float4 t = v[j];
            t.x = (t.x - mean) * rstd; t.y = (t.y - mean) * rstd; t.z = (t.z - mean) * rstd; t.w = (t.w - mean) * rstd;
            if (w) { const float4 ww = ((const float4*)w)[i]; t.x *= ww.x + wadd; t.y *= ww.y + wadd; t.z *= ww.z + wadd; t.w *= ww.w + wadd; }
            if (b) { const float4 bb = ((const float4*)b)[i]; t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w; }
            h[0] = (_Float16)t.x; h[1] = (_Float16)t.y; h[2] = (_Float16)t.z; h[3] = (_Float16)t.w;
        }
        *(half4_t*)(yr + i * 4) = h;
    }
}
// R rows per wave (round 5): the loads of all R rows are issued before the first reduction, so a wave keeps R x 1.25 .. 5 KB in flight instead of one row's —
// k_layer_norm_f16 ran at 40 % of the HBM peak with traffic = algorithmic bytes (profiles/r05a_pmc_traffic_layer_norm_f16.json): latency, not bytes.  The arithmetic
// of a row is the single-row kernel's, statement for statement (bit-identical outputs).  MEASURED SLOWER and left off (option "ln16_rows" = 4 selects it): see g_ln16_rows.
template <int NV, int R>
__global__ __launch_bounds__(256) void k_layer_norm_f16_rows(_Float16* __restrict__ dst, const float* __restrict__ x, int ne0, int Kp, int64_t nrows, int64_t xs,
                                                             float eps, const float* __restrict__ w, const float* __restrict__ b, int rms, int64_t mod_L) {
    const int lane     = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= nrows) return;
    const float wadd = mod_L > 0 ? 1.f : 0.f;
    const int n4     = ne0 / 4;
    float4 v[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r < nrows ? row0 + r : nrows - 1;  // a ragged last group re-reads the last row (its stores are masked below)
        const float4* xr  = (const float4*)(x + row * xs);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            v[r][j]     = i < n4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = 0.f;
        if (!rms) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) s += (v[r][j].x + v[r][j].y) + (v[r][j].z + v[r][j].w);
            mean[r] = wave_sum(s) / (float)ne0;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (lane + 64 * j < n4) {
                const float a = v[r][j].x - mean[r], bb = v[r][j].y - mean[r], c = v[r][j].z - mean[r], d = v[r][j].w - mean[r];
                q += (a * a + bb * bb) + (c * c + d * d);
            }
        }
        rstd[r] = rsqrtf(wave_sum(q) / (float)ne0 + eps);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r;
        if (row >= nrows) break;
        const float* wr = w;
        const float* br = b;
        if (mod_L > 0) {
            wr = w + (row / mod_L) * ne0;
            br = b + (row / mod_L) * ne0;
        }
        _Float16* yr = dst + row * Kp;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i >= Kp / 4) continue;
            half4_t h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (i < n4) {
                float4 t = v[r][j];
                t.x = (t.x - mean[r]) * rstd[r]; t.y = (t.y - mean[r]) * rstd[r]; t.z = (t.z - mean[r]) * rstd[r]; t.w = (t.w - mean[r]) * rstd[r];
                if (wr) { const float4 ww = ((const float4*)wr)[i]; t.x *= ww.x + wadd; t.y *= ww.y + wadd; t.z *= ww.z + wadd; t.w *= ww.w + wadd; }
                if (br) { const float4 bb = ((const float4*)br)[i]; t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w; }
                h[0] = (_Float16)t.x; h[1] = (_Float16)t.y; h[2] = (_Float16)t.z; h[3] = (_Float16)t.w;
            }
            *(half4_t*)(yr + i * 4) = h;
        }
    }
}
static int g_ln16_rows = 1;  // option "ln16_rows": rows per wave of the register-resident LayerNorm -> f16 image kernel.  Default 1 = the single-row kernel: the multi-row form (4) measured SLOWER — SD1.5 family 0.963 -> 1.032 ms, SDXL 0.343 -> 0.381 ms, step 21.98 -> 22.08 ms (profiles/r06e_*): fewer, fatter waves hide less than many thin ones here
void gemm16_set_ln16_rows(int v) { g_ln16_rows = v; }
void launch_layer_norm_f16(hipStream_t s, void* dst, const float* x, int64_t ne0, int64_t nrows, int64_t xs, float eps, const float* w, const float* b, bool rms,
                           int64_t mod_L) {
    KScope ks_(s, KF_LN_F16, 0.0, (double)nrows * ne0 * 4.0 + (double)nrows * rup64(ne0, 64) * 2.0);
    const int Kp = (int)rup64(ne0, 64);
#define LN16_ARGS (_Float16*)dst, x, (int)ne0, Kp, nrows, xs, eps, w, b, rms ? 1 : 0, mod_L
    const unsigned grid = (unsigned)((nrows + 3) / 4);
    if (g_ln16_rows > 1 && nrows >= 4096 && ne0 % 4 == 0) {  // enough rows to fill the chip with multi-row waves
        if (Kp <= 256 * 2) return (void)k_layer_norm_f16_rows<2, 4><<<(unsigned)((nrows + 15) / 16), 256, 0, s>>>(LN16_ARGS);
        if (Kp <= 256 * 5) return (void)k_layer_norm_f16_rows<5, 2><<<(unsigned)((nrows + 7) / 8), 256, 0, s>>>(LN16_ARGS);
    }
    if (Kp <= 256 * 2)
        k_layer_norm_f16_reg<2><<<grid, 256, 0, s>>>(LN16_ARGS);
    else if (Kp <= 256 * 5)
        k_layer_norm_f16_reg<5><<<grid, 256, 0, s>>>(LN16_ARGS);
    else if (Kp <= 256 * 12)
        k_layer_norm_f16_reg<12><<<grid, 256, 0, s>>>(LN16_ARGS);
    else
        k_layer_norm_f16<<<grid, 256, 0, s>>>(LN16_ARGS);
#undef LN16_ARGS
}

// GEGLU writing the f16 operand image: dst[t][i] = x[t][i] * gelu(x[t][inner+i])
__global__ void k_geglu_f16(_Float16* __restrict__ dst, const float* __restrict__ x, int64_t tokens, int inner, int Kp, int64_t xs) {
    const int64_t n4 = tokens * (Kp / 4);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / (Kp / 4);
        const int c     = (int)(i - t * (Kp / 4)) * 4;
        half4_t h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (c < inner) {
            const float4 a = *(const float4*)(x + t * xs + c), gt = *(const float4*)(x + t * xs + inner + c);
            h[0] = (_Float16)(a.x * act_apply<UN_GELU>(gt.x));
            h[1] = (_Float16)(a.y * act_apply<UN_GELU>(gt.y));
            h[2] = (_Float16)(a.z * act_apply<UN_GELU>(gt.z));
            h[3] = (_Float16)(a.w * act_apply<UN_GELU>(gt.w));
        }
        *(half4_t*)(dst + t * Kp + c) = h;
    }
}
void launch_geglu_f16(hipStream_t s, void* dst, const float* x, int64_t tokens, int64_t inner, int64_t xs) {
    KScope ks_(s, KF_PACK_F16, 0.0, (double)tokens * inner * 8.0 + (double)tokens * rup64(inner, 64) * 2.0);
    const int Kp = (int)rup64(inner, 64);
    const int64_t n4 = tokens * (Kp / 4);
    int64_t blocks   = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    k_geglu_f16<<<(unsigned)blocks, 256, 0, s>>>((_Float16*)dst, x, tokens, (int)inner, Kp, xs);
}

// GroupNorm statistics -> per-(image, channel) affine:  y = x * scale[n][c] + shift[n][c]
template <int NT>
__global__ __launch_bounds__(NT) void k_gn_stats(float* __restrict__ scale, float* __restrict__ shift, const float* __restrict__ x, int64_t hw, int C, int groups,
                                                 int cpg, float eps, const float* __restrict__ w, const float* __restrict__ b) {
    __shared__ float scratch[2 * (NT / 64)];
    const int gidx = blockIdx.x % groups, n = blockIdx.x / groups;
    const int c0 = gidx * cpg, c1 = min(c0 + cpg, C);
    if (c0 >= c1) return;
    const int64_t cnt = (int64_t)(c1 - c0) * hw;
    const float* xs   = x + ((int64_t)n * C + c0) * hw;
    const bool v4     = (hw % 4 == 0) && ((((uintptr_t)xs) & 15) == 0);
    float s = 0.f, dummy = 0.f;
    if (v4) {
        for (int64_t i = threadIdx.x; i < cnt / 4; i += NT) {
            const float4 v = ((const float4*)xs)[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int64_t i = threadIdx.x; i < cnt; i += NT) s += xs[i];
    }
    block_sum2<NT / 64>(s, dummy, scratch);
    const float mean = s / (float)cnt;
    float q = 0.f;
    if (v4) {
        for (int64_t i = threadIdx.x; i < cnt / 4; i += NT) {
            const float4 v = ((const float4*)xs)[i];
            const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    } else {
        for (int64_t i = threadIdx.x; i < cnt; i += NT) {
            const float a = xs[i] - mean;
            q += a * a;
        }
    }
    dummy = 0.f;
    block_sum2<NT / 64>(q, dummy, scratch);
    const float rstd = rsqrtf(q / (float)cnt + eps);
    for (int c = c0 + threadIdx.x; c < c1; c += NT) {
        const float sc           = (w ? w[c] : 1.f) * rstd;
        scale[(int64_t)n * C + c] = sc;
        shift[(int64_t)n * C + c] = (b ? b[c] : 0.f) - mean * sc;
    }
}
// TWO-SOURCE form of the statistics / apply kernels (round 4): the activation is the channel concatenation [x (C1 channels) | x2 (C - C1 channels)] of two NCHW
// tensors that is never materialised (UNet skip connections: CONCAT(h, skip) read only by a GroupNorm and by the skip 1x1 conv, unet.hpp:702).  Float4 i4 of the
// (image n, channels c0..) slab: hw % 4 == 0, so the four elements share their channel and their source.  x2 == nullptr: the single contiguous tensor.
__device__ __forceinline__ float4 gn_ld4(const float* __restrict__ x, const float* __restrict__ x2, int C1, int C, int n, int c0, int64_t hw, int64_t i4) {
    if (!x2) return ((const float4*)(x + ((int64_t)n * C + c0) * hw))[i4];
    const int64_t e  = i4 * 4;
    const int ch     = c0 + (int)(e / hw);
    const int64_t of = e - (int64_t)(ch - c0) * hw;
    const float* p   = ch < C1 ? x + ((int64_t)n * C1 + ch) * hw : x2 + ((int64_t)n * (C - C1) + (ch - C1)) * hw;
    return *(const float4*)(p + of);
}
// register-resident variant: one (image, group) slab of up to 4 * NT * NV floats is read ONCE into NV float4 registers per thread
// (the kernel above reads it twice: mean, then centred sum of squares); same per-thread order, same block reductions
template <int NT, int NV>
__global__ __launch_bounds__(NT) void k_gn_stats_reg(float* __restrict__ scale, float* __restrict__ shift, const float* __restrict__ x, int64_t hw, int C, int groups,
                                                     int cpg, float eps, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ x2 = nullptr, int C1 = 0) {
    __shared__ float scratch[2 * (NT / 64)];
    const int gidx = blockIdx.x % groups, n = blockIdx.x / groups;
    const int c0 = gidx * cpg, c1 = min(c0 + cpg, C);
    if (c0 >= c1) return;
    const int64_t cnt = (int64_t)(c1 - c0) * hw;
    const int64_t n4  = cnt / 4;
    float4 v[NV];
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t i = threadIdx.x + (int64_t)NT * j;
        v[j]            = i < n4 ? gn_ld4(x, x2, C1, C, n, c0, hw, i) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    block_sum2<NT / 64>(s, dummy, scratch);
    const float mean = s / (float)cnt;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (threadIdx.x + (int64_t)NT * j < n4) {
            const float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    dummy = 0.f;
    block_sum2<NT / 64>(q, dummy, scratch);
    const float rstd = rsqrtf(q / (float)cnt + eps);
    for (int c = c0 + threadIdx.x; c < c1; c += NT) {
        const float sc            = (w ? w[c] : 1.f) * rstd;
        scale[(int64_t)n * C + c] = sc;
        shift[(int64_t)n * C + c] = (b ? b[c] : 0.f) - mean * sc;
    }
}
// groups too large for registers (the KL-VAE's 128-channel 512x512 maps: 4 channels x 262144 pixels per group): ONE pass with sums
// taken relative to the group's first element (shifted data: var = E[(x-K)^2] - E[x-K]^2 stays well conditioned because K is a sample of
// the group; the two-pass kernel reads every byte twice and ran at 2.7 TB/s algorithmic)
template <int NT>
__global__ __launch_bounds__(NT) void k_gn_stats_1pass(float* __restrict__ scale, float* __restrict__ shift, const float* __restrict__ x, int64_t hw, int C, int groups,
                                                       int cpg, float eps, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ x2 = nullptr, int C1 = 0) {
    __shared__ float scratch[2 * (NT / 64)];
    const int gidx = blockIdx.x % groups, n = blockIdx.x / groups;
    const int c0 = gidx * cpg, c1 = min(c0 + cpg, C);
    if (c0 >= c1) return;
    const int64_t cnt = (int64_t)(c1 - c0) * hw;
    const float K     = gn_ld4(x, x2, C1, C, n, c0, hw, 0).x;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = threadIdx.x; i < cnt / 4; i += NT) {
        const float4 v = gn_ld4(x, x2, C1, C, n, c0, hw, i);
        const float a = v.x - K, bb = v.y - K, c = v.z - K, d = v.w - K;
        s1 += (a + bb) + (c + d);
        s2 += (a * a + bb * bb) + (c * c + d * d);
    }
    block_sum2<NT / 64>(s1, s2, scratch);
    const float m1   = s1 / (float)cnt;
    const float mean = K + m1;
    const float var  = fmaxf(s2 / (float)cnt - m1 * m1, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (int c = c0 + threadIdx.x; c < c1; c += NT) {
        const float sc            = (w ? w[c] : 1.f) * rstd;
        scale[(int64_t)n * C + c] = sc;
        shift[(int64_t)n * C + c] = (b ? b[c] : 0.f) - mean * sc;
    }
}
// FEW (image, group) slabs, each large (the KL-VAE at batch 1: 32 slabs of 16.8 MB at the 512 x 512 x 128 level; SDXL's first UNet level at batch 1):
// one workgroup per slab leaves 7/8 of the chip idle — 30 launches = 8.0 ms of the 25 ms 1024 x 1024 decode at 0.9 TB/s (profiles/r07i_vae_families.txt).
// Here P workgroups share a slab: each sums its contiguous part relative to the slab's first element K (same shifted-data form as k_gn_stats_1pass)
// into part[(slab * P + p) * 2 + {0, 1}]; k_gn_stats_final adds the P partial pairs in a fixed order (bitwise deterministic) and writes scale / shift.
template <int NT>
__global__ __launch_bounds__(NT) void k_gn_stats_part(float* __restrict__ part, const float* __restrict__ x, int64_t hw, int C, int groups, int cpg, int P) {
    __shared__ float scratch[2 * (NT / 64)];
    const int p = blockIdx.x % P, slab = blockIdx.x / P;
    const int gidx = slab % groups, n = slab / groups;
    const int c0 = gidx * cpg, c1 = min(c0 + cpg, C);
    if (c0 >= c1) return;
    const int64_t n4 = (int64_t)(c1 - c0) * hw / 4;
    const float4* xs = (const float4*)(x + ((int64_t)n * C + c0) * hw);
    const float K    = ((const float*)xs)[0];
    const int64_t i0 = n4 * p / P, i1 = n4 * (p + 1) / P;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += NT) {
        const float4 v = xs[i];
        const float a = v.x - K, bb = v.y - K, c = v.z - K, d = v.w - K;
        s1 += (a + bb) + (c + d);
        s2 += (a * a + bb * bb) + (c * c + d * d);
    }
    block_sum2<NT / 64>(s1, s2, scratch);
    if (threadIdx.x == 0) {
        part[(int64_t)blockIdx.x * 2]     = s1;
        part[(int64_t)blockIdx.x * 2 + 1] = s2;
    }
}
__global__ __launch_bounds__(64) void k_gn_stats_final(float* __restrict__ scale, float* __restrict__ shift, const float* __restrict__ part, const float* __restrict__ x,
                                                       int64_t hw, int C, int groups, int cpg, int P, float eps, const float* __restrict__ w, const float* __restrict__ b) {
    const int slab = blockIdx.x, gidx = slab % groups, n = slab / groups;
    const int c0 = gidx * cpg, c1 = min(c0 + cpg, C);
    if (c0 >= c1) return;
    const int64_t cnt = (int64_t)(c1 - c0) * hw;
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < P; ++p) {  // every lane the same fixed order
        s1 += part[((int64_t)slab * P + p) * 2];
        s2 += part[((int64_t)slab * P + p) * 2 + 1];
    }
    const float K    = x[((int64_t)n * C + c0) * hw];
    const float m1   = s1 / (float)cnt;
    const float mean = K + m1;
    const float var  = fmaxf(s2 / (float)cnt - m1 * m1, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (int c = c0 + threadIdx.x; c < c1; c += 64) {
        const float sc            = (w ? w[c] : 1.f) * rstd;
        scale[(int64_t)n * C + c] = sc;
        shift[(int64_t)n * C + c] = (b ? b[c] : 0.f) - mean * sc;
    }
}
static int64_t g_gn_split_min = 1 << 16;  // option "gn_split_min": least floats per (image, group) slab for the shared-slab form (0 turns it off: 1 << 62)
void gemm16_set_gn_split_min(int v) { g_gn_split_min = v <= 0 ? (1ll << 62) : (int64_t)v; }
// workgroups per slab for that form (0: the one-workgroup-per-slab kernels); the caller provides N * groups * P * 2 floats of scratch
int gn_stats_split(int64_t hw, int64_t C, int64_t N, int groups) {
    const int64_t cpg = (C + groups - 1) / groups, cnt = cpg * hw, slabs = N * groups;
    if (hw % 4 != 0 || C % groups != 0 || cnt < g_gn_split_min || slabs >= 128) return 0;  // from 256 KB per slab on, fewer than 128 slabs
    int64_t P = 512 / slabs;
    if (P > 32) P = 32;
    while (P > 1 && cnt / 4 / P < 2048) P /= 2;  // at least 32 KB per workgroup
    return P >= 2 ? (int)P : 0;
}

// x2 != nullptr: statistics of the channel concatenation [x (C1 channels) | x2] (gn_two_source_supported)
bool gn_two_source_supported(const float* x, const float* x2, int64_t hw, int64_t C, int64_t C1, int groups) {
    const int64_t cnt = ((C + groups - 1) / groups) * hw;
    return hw % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)x2)) & 15) == 0 && C1 > 0 && C1 < C && (cnt <= 4 * 1024 * 16 || cnt >= 16384) && C * hw < (1ll << 31);
}
void launch_gn_stats(hipStream_t s, float* scale, float* shift, const float* x, int64_t hw, int64_t C, int64_t N, int groups, float eps, const float* w,
                     const float* b, const float* x2, int64_t C1, float* part) {
    KScope ks_(s, KF_GN_STATS, 0.0, (double)hw * C * N * 4.0);  // algorithmic: ONE read of the activation
    const int cpg       = (int)((C + groups - 1) / groups);
    const int64_t cnt   = (int64_t)cpg * hw;
    const bool v4       = hw % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)x2)) & 15) == 0;  // (a short last group is fine: the kernels bound it by c1)
    const unsigned grid = (unsigned)(N * groups);
    if (part && !x2 && v4) {
        if (const int P = gn_stats_split(hw, C, N, groups)) {
            k_gn_stats_part<1024><<<grid * (unsigned)P, 1024, 0, s>>>(part, x, hw, (int)C, groups, cpg, P);
            k_gn_stats_final<<<grid, 64, 0, s>>>(scale, shift, part, x, hw, (int)C, groups, cpg, P, eps, w, b);
            return;
        }
    }
    if (x2 && !gn_two_source_supported(x, x2, hw, C, C1, groups)) {
        fprintf(stderr, "ggml-mi355x: two-source GroupNorm statistics asked for a shape they do not take\n");
        abort();
    }
    if (v4 && cnt <= 4 * 256 * 4)
        k_gn_stats_reg<256, 4><<<grid, 256, 0, s>>>(scale, shift, x, hw, (int)C, groups, cpg, eps, w, b, x2, (int)C1);
    else if (v4 && cnt <= 4 * 1024 * 4)
        k_gn_stats_reg<1024, 4><<<grid, 1024, 0, s>>>(scale, shift, x, hw, (int)C, groups, cpg, eps, w, b, x2, (int)C1);
    else if (v4 && cnt <= 4 * 1024 * 16)
        k_gn_stats_reg<1024, 16><<<grid, 1024, 0, s>>>(scale, shift, x, hw, (int)C, groups, cpg, eps, w, b, x2, (int)C1);
    else if (v4 && cnt >= 16384)
        k_gn_stats_1pass<1024><<<grid, 1024, 0, s>>>(scale, shift, x, hw, (int)C, groups, cpg, eps, w, b, x2, (int)C1);
    else if (cnt >= 16384)
        k_gn_stats<1024><<<grid, 1024, 0, s>>>(scale, shift, x, hw, (int)C, groups, cpg, eps, w, b);
    else
        k_gn_stats<256><<<grid, 256, 0, s>>>(scale, shift, x, hw, (int)C, groups, cpg, eps, w, b);
}

// f32 NCHW [hw][C][N] -> f16 NHWC [N][hw][Cp] with optional per-(n,c) affine (GroupNorm apply) and SiLU.
// 64 channels x 64 positions per workgroup through an LDS transpose: coalesced 256-B reads along hw, 128-B writes along c.
__global__ __launch_bounds__(256) void k_nchw_to_nhwc_f16(_Float16* __restrict__ dst, const float* __restrict__ x, int64_t hw, int C, int Cp,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int silu, float post_mul, float* dst_f32) {
    __shared__ float tile[64][65];
    const int n  = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* xn = x + (int64_t)n * C * hw;
    for (int j = ty; j < 64; j += 4) {
        const int c = c0 + j, p = p0 + tx;
        float v = 0.f;
        if (c < C && p < hw) {
            v = xn[(int64_t)c * hw + p];
            if (scale) v = v * scale[(int64_t)n * C + c] + shift[(int64_t)n * C + c];
            if (silu == 1) v = act_apply<UN_SILU>(v);
            else if (silu == 2) v = act_apply<UN_RELU>(v);
            if (dst_f32) dst_f32[((int64_t)n * C + c) * hw + p] = v;  // (x may alias dst_f32: every element is read and written by the same thread)
            v *= post_mul;
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    _Float16* dn = dst + (int64_t)n * hw * Cp;
    for (int j = ty; j < 64; j += 4) {
        const int p = p0 + j, c = c0 + tx;
        if (p < hw && c < Cp) dn[(int64_t)p * Cp + c] = (_Float16)tile[tx][j];
    }
}
// 16-byte variant (hw % 4 == 0, 16-byte aligned rows): 64 channels x 64 positions per workgroup.  Load: a thread owns a channel PAIR x 4
// consecutive positions — two float4 loads (256-byte row segments per 16 lanes), affine + SiLU, four half2 LDS stores (row stride 33
// dwords: 2-way conflicts at most).  Store: a thread owns one position x 8 channels — one 16-byte store (128-byte rows per 8 lanes).
__global__ __launch_bounds__(256) void k_nchw_to_nhwc_f16_v4(_Float16* __restrict__ dst, const float* __restrict__ x, int64_t hw, int C, int Cp,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, int silu,
                                                             const float* __restrict__ x2 = nullptr, int C1 = 0, _Float16* __restrict__ dst_raw = nullptr,
                                                             float post_mul = 1.f, float* dst_f32 = nullptr) {
    __shared__ uint32_t tile[64][33];  // [position][channel pair] half2
    __shared__ uint32_t tile_raw[64][33];  // dst_raw != nullptr: the same values WITHOUT affine / SiLU (the image the skip 1x1 conv reads)
    const int n  = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    // channel row pointer: one tensor, or the concatenation [x (C1 channels) | x2 (C - C1)] (gn_ld4)
    auto rowp = [&](int c) -> const float* {
        if (!x2) return x + ((int64_t)n * C + c) * hw;
        return c < C1 ? x + ((int64_t)n * C1 + c) * hw : x2 + ((int64_t)n * (C - C1) + (c - C1)) * hw;
    };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = threadIdx.x + 256 * k;  // 32 channel pairs x 16 position quads
        const int cp = item >> 4, pq = (item & 15) * 4;
        const int c = c0 + 2 * cp, p = p0 + pq;
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        if (p < hw) {
            if (c < C) va = *(const float4*)(rowp(c) + p);
            if (c + 1 < C) vb = *(const float4*)(rowp(c + 1) + p);
        }
        float a[4] = {va.x, va.y, va.z, va.w}, bq[4] = {vb.x, vb.y, vb.z, vb.w};
        if (dst_raw) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 hu = (_Float16)(c < C ? a[i] : 0.f), hv = (_Float16)(c + 1 < C ? bq[i] : 0.f);
                tile_raw[pq + i][cp] = (uint32_t)__builtin_bit_cast(uint16_t, hu) | ((uint32_t)__builtin_bit_cast(uint16_t, hv) << 16);
            }
        }
        if (scale) {
            const float sa = c < C ? scale[(int64_t)n * C + c] : 0.f, ha = c < C ? shift[(int64_t)n * C + c] : 0.f;
            const float sb = c + 1 < C ? scale[(int64_t)n * C + c + 1] : 0.f, hb = c + 1 < C ? shift[(int64_t)n * C + c + 1] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i]  = a[i] * sa + ha;
                bq[i] = bq[i] * sb + hb;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float u = a[i], v = bq[i];
            if (silu == 1) {
                u = act_apply<UN_SILU>(u);
                v = act_apply<UN_SILU>(v);
            } else if (silu == 2) {  // ReLU in front of a conv (TAESD, tae.hpp:15-76)
                u = act_apply<UN_RELU>(u);
                v = act_apply<UN_RELU>(v);
            }
            a[i] = u, bq[i] = v;  // (kept for the f32 write-back below)
            u *= post_mul;  // Conv2d scale (ggml_ext_conv_2d: x = scale(x, s) before the f16 im2col) folded into the operand image
            v *= post_mul;
            if (c >= C) u = 0.f;       // padded channels of the operand image are zeros
            if (c + 1 >= C) v = 0.f;
            const _Float16 hu = (_Float16)u, hv = (_Float16)v;
            tile[pq + i][cp] = (uint32_t)__builtin_bit_cast(uint16_t, hu) | ((uint32_t)__builtin_bit_cast(uint16_t, hv) << 16);
        }
        if (dst_f32 && p < hw) {  // the activated values as f32 NCHW too (in place over x: this thread read exactly these eight floats)
            if (c < C) *(float4*)(dst_f32 + ((int64_t)n * C + c) * hw + p) = make_float4(a[0], a[1], a[2], a[3]);
            if (c + 1 < C) *(float4*)(dst_f32 + ((int64_t)n * C + c + 1) * hw + p) = make_float4(bq[0], bq[1], bq[2], bq[3]);
        }
    }
    __syncthreads();
    _Float16* dn = dst + (int64_t)n * hw * Cp;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = threadIdx.x + 256 * k;  // 64 positions x 8 channel octets
        const int pos = item >> 3, cg = (item & 7) * 4;  // cg: first channel pair of the octet
        const int p = p0 + pos, c = c0 + 2 * cg;
        if (p < hw && c < Cp) {
            uint4 o;
            o.x = tile[pos][cg];
            o.y = tile[pos][cg + 1];
            o.z = tile[pos][cg + 2];
            o.w = tile[pos][cg + 3];
            *(uint4*)(dn + (int64_t)p * Cp + c) = o;
            if (dst_raw) {
                uint4 r;
                r.x = tile_raw[pos][cg];
                r.y = tile_raw[pos][cg + 1];
                r.z = tile_raw[pos][cg + 2];
                r.w = tile_raw[pos][cg + 3];
                *(uint4*)(dst_raw + (int64_t)n * hw * Cp + (int64_t)p * Cp + c) = r;
            }
        }
    }
}
// x2 != nullptr: the source is the channel concatenation [x (C1 channels) | x2 (C - C1)] (both NCHW, never materialised);  dst_raw != nullptr: a second
// image of the same values without affine / SiLU (one read of the sources for the GroupNorm'ed conv operand AND the skip 1x1 conv's operand)
void launch_nchw_to_nhwc_f16(hipStream_t s, void* dst, const float* x, int64_t hw, int64_t C, int64_t N, const float* scale, const float* shift, int silu,
                             const float* x2, int64_t C1, void* dst_raw, float post_mul, float* dst_f32) {
    KScope ks_(s, KF_NCHW_NHWC, 0.0, (double)hw * C * N * 4.0 * (dst_f32 ? 2.0 : 1.0) + (double)hw * rup64(C, 64) * N * 2.0 * (dst_raw ? 2.0 : 1.0));
    if (dst_f32 && x2) {
        fprintf(stderr, "ggml-mi355x: NCHW -> NHWC pass with an f32 write-back takes one source\n");
        abort();
    }
    const int Cp = (int)rup64(C, 64);
    dim3 grid((unsigned)((hw + 63) / 64), (unsigned)(Cp / 64), (unsigned)N);
    const bool v4 = hw % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)dst) | ((uintptr_t)x2) | ((uintptr_t)dst_raw) | ((uintptr_t)dst_f32)) & 15) == 0;
    if ((x2 || dst_raw) && !v4) {
        fprintf(stderr, "ggml-mi355x: two-source / two-output NCHW -> NHWC pass needs hw %% 4 == 0 and 16-byte aligned tensors\n");
        abort();
    }
    if (v4)
        k_nchw_to_nhwc_f16_v4<<<grid, 256, 0, s>>>((_Float16*)dst, x, hw, (int)C, Cp, scale, shift, silu, x2, (int)C1, (_Float16*)dst_raw, post_mul, dst_f32);
    else
        k_nchw_to_nhwc_f16<<<grid, 256, 0, s>>>((_Float16*)dst, x, hw, (int)C, Cp, scale, shift, silu, post_mul, dst_f32);
}

}  // namespace mi355x
