// flash_attn.hip — fused scaled-dot-product attention forward for gfx950 (no mask, no GQA broadcast needed on
// the UNet / VAE path): softmax(scale * Q K^T) V with online softmax, never materialising the [Lk, Lq] scores.
// Serves both encodings the reference emits (ggml_ext_attention_ext, src/core/ggml_extend.hpp:1349-1485):
//   * the FLASH_ATTN_EXT node (flash flag on; K/V arrive as f16), and
//   * the manual MUL_MAT -> SCALE -> SOFT_MAX -> MUL_MAT chain (flag off; K/V f32), pattern-matched by the planner.
// Numerics: Q, K, P and V enter v_mfma_f32_32x32x16_f16 as f16, all sums and the softmax are f32 — tighter than
// ggml-cpu's flash path (which accumulates V in f16).
//
// Mapping (wave64): a workgroup = 4 waves x 32 query rows for one (head, image); K/V tiles of 64 keys are staged
// in LDS (K row-major [key][d], V TRANSPOSED [d][key] so the PV B-fragments are k-contiguous 8-byte reads).
// S^T = K.Q^T is computed "swapped" (MFMA A = K rows, B = Q rows): every lane then holds the scores of ONE query
// (its MFMA column) for 32 keys, its partner lane (l ^ 32) the other 32, so the row max / row sum need a single
// cross-lane exchange.  P is fed back as the A operand WITHOUT any data movement by letting the MFMA k-slots
// follow the accumulator's own key order (lane half h owns keys {4h..4h+3, 8+4h..8+4h+3} of each 16-key group)
// and reading V with the same permutation.
// Staging is split (cdna_hip_programming.md T14): the 16-byte global loads of tile t+1 are issued into registers
// BEFORE the MFMAs of tile t and written to LDS after the next barrier, so HBM/L2 latency hides under compute.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"
#include "ktime.h"

namespace mi355x {

constexpr int FA_KT  = 64;  // keys per tile
constexpr float FA_THR = 8.0f;  // deferred-max threshold, log2 units
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
constexpr int FA_VTS = 68;  // Vt row stride in halfs (136 B: conflict-free ds_read_b64 over 32 rows)

struct FAArgs {
    const char *q, *k, *v;
    float* dst;
    _Float16* dst16;
    int64_t q_nb1, q_nb2, k_nb1, k_nb2, v_nb0, v_nb1, v_nb2, dst_nb_q, dst_nb_h, dst_nb_n, ld16;
    int H;
    int Lq, Lk, D, DV;
    int kv_f16;
    int grp, units;  // XCD-grouped 1-D grid (see the kernel): heads per unit, (image, query block) units
    int q_f16;   // Q is an f16 image (planner: the projection GEMM's head-major output feeds only this launch)
    int q_vec;   // Q rows are f32 d-contiguous, 16-byte aligned, D % 4 == 0 -> coalesced float4 staging through LDS
    int vec_ok;  // K and V rows are d-contiguous, 16-byte aligned, D % 8 == 0 -> 128-bit staging loads
    float scale_log2e;
    int qi;      // k_flash_short: 32-query blocks per wave
};

// FAST: K and V are f16, d-contiguous and 16-byte aligned (the FLASH_ATTN_EXT node as the reference builds it) -> 128-bit
// loads, register-prefetched one tile ahead.  !FAST: any strides / f32 sources (manual-attention chain), staged directly.
// ABL != 0: TIMING ABLATIONS with wrong results (option "flash_ablate", scripts/flash_ablation.py only): 1 = no softmax VALU work (the raw
// scores go into the PV product), 2 = K/V tiles are staged once and reused (no global loads / LDS stores in the loop)
// MSLOT (d = 40 on the 48-wide tile, FAST): the running max rides in the first padded k-slot — Q[q][40] = -m_run[q], K[key][40] = 1 — so the MFMA
// delivers scores already relative to the max and the 32 v_sub per lane per tile disappear from the VALU-bound loop (m_run is kept
// f16-representable; any consistent offset is a valid softmax shift because numerator and row sum use the same P)
// Waves per SIMD the register allocator aims for at head dims <= 64 (the second __launch_bounds__ argument is waves per SIMD, and it is what
// decides the residency: with 3 the compiler spends 160 VGPRs and three workgroups share a CU — measured SQ_WAVE_CYCLES / SIMD cycles = 2.67, the
// 16 workgroups per CU of the L = 4096 launch run as 3+3+3+3+3+1 — although the 31.7 KB of LDS would admit five).  Compiled for 4 the kernel
// spills ~25 registers into the tile loop and is 20-27 % SLOWER (5.6-6.1 vs 4.6-4.8 ms per SD1.5 forward); running the softmax over two 32-key
// halves (16 live score registers instead of 32) did not remove the spills and changed nothing at 3 (profiles/r03e_flash_occupancy.txt).
// Getting to 4 waves per SIMD needs a hand-made register budget (index arrays recomputed per tile, V fragments not hoisted), not a flag.
#ifndef FA_OCC_SMALL
#define FA_OCC_SMALL 3
#endif

// QB = query blocks (32 rows each) per wave.  QB = 2 (FAST only; round 3): a wave owns 64 queries, a workgroup 256 — every K / V fragment read
// from LDS, every staged K / V byte and every barrier then feeds TWO MFMAs instead of one, and the two blocks' QK^T -> softmax -> PV chains are
// independent instruction streams the scheduler can interleave (the round-2 kernel was a single dependent chain per wave: MFMA pipe 27-30 % busy,
// 42 % of the wave cycles waiting, profiles/r03c_pmc_flash.txt).  Costs 2x the accumulator / score registers: 2 waves per SIMD.
// VPF: fragment reads issued AHEAD of the MFMAs that use them — the first V fragments of a tile before its softmax VALU work, the rest
// four fragments ahead inside the PV loop; for d >= 128 the K fragments of QK^T in groups of four, one group ahead.  Left to the compiler every
// MFMA (pair) was preceded by its own ds_read + s_waitcnt lgkmcnt: the LDS latency (~100 cycles) sat in front of each of the 16-32 MFMAs of a tile.
// Measured (profiles/r04z_flash_vpf.txt, one box, alternating variants): d = 128, L = 4352: 409 -> 387 us; d = 64, L = 4250: 598 -> 553; d = 64,
// L = 4096: 154 -> 148; d = 80: 99.5 -> 96.0; d = 40 with two query blocks: 641 -> 637.  Option "flash_vpf" (bit per head-dim class) for A/B runs.
// (Also measured, rejected and removed: a software-pipelined tile loop computing S(t+1) = K(t+1) Q^T inside the exponentials of tile t — one
// scheduling region, MFMAs and VALU interleaved by sched_group_barrier, bit-identical results: 15-25 % SLOWER at d <= 64 (second score register set:
// 3 -> 2 waves per SIMD), equal at d = 128: profiles/r04x_flash_sp_rejected.txt.)
// VTR (FAST only; option "flash_vtr"): V tiles stay ROW-MAJOR [key][dv] in LDS — staged with the same coalesced 16-byte loads and 16-byte LDS
// writes as K — and the PV B-fragments are fetched with the gfx950 transposing read (ds_read_b64_tr_b16), instead of transposing in the staging
// pass (8 two-byte LDS writes per chunk: at d = 128 that is 32 ds_write_b16 per thread per tile next to 32 MFMAs, fed by loads that touch 64
// different rows per instruction).  The read (cdna_hip_programming.md, LDS): inside each 16-lane group, lane i supplies the address of 4
// consecutive halfs; result element j of lane l is half (l & 3) of the 8 bytes addressed by lane 4 j + ((l & 15) >> 2) of the same group.  With
// lane i pointing at V[k0 + (i >> 2)][n0 + 4 (i & 3) ..], lane l receives V[k0 + j][n0 + (l & 15)], j = 0..3: four consecutive keys of one
// output column — half a B fragment (the other half is the same read 8 keys further on).
typedef short fa_short4_t __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ half4_t lds_read_tr16(const _Float16* p) {
    const fa_short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_short4_t*)p);
    return __builtin_bit_cast(half4_t, v);
}
// row stride (halfs) of the row-major V tile: a 32-lane half of the transposing read covers 4 keys x 32 columns (64 bytes per key), so rows
// 16 dwords apart modulo the 64 banks never collide
constexpr int fa_vtr_stride(int ndv) {
    int dw = ndv * 16;
    if (dw % 64 != 16 && dw % 64 != 48) dw += 16;
    return dw * 2;
}
// OVL (two query blocks, d <= 48, VPF; option "flash_ovl"): the tile is issued so that one block's softmax VALU work sits in the issue gaps of the OTHER block's MFMAs, inside one scheduling
// region each:   QK(b0) | max(b0) | { QK(b1) || exp, cvt (b0) } | max(b1) | { PV(b0) || exp, cvt (b1) } | PV(b1)
// (a 32x32x16 MFMA occupies its SIMD's matrix pipe for 32 cycles and hides up to ~5 single-issue VALU / LDS instructions of the same wave in that
// gap, MI355X_MICROARCH.md; the default stream runs QK -> softmax -> QK -> softmax -> PV as separate phases: 28 MFMAs = 900 cycles and ~160 VALU
// of a tile one after the other).  Same arithmetic per query in the same order: bit-identical results.  Costs: both blocks' score registers are
// live during region A, and the tile's eight V fragments stay in registers for the second P V pass.  Measured (profiles/r04L_flash_ovl.txt, alternating
// variants, bit-identical outputs): d = 40, L = 4096: 601 -> 582 us; Lq = 2048, Lk = 1000 (ragged last tile): 111 -> 105 us.  The gain is small because
// the kernel is bound by its instruction ISSUE (one v_exp_f32 per score), not by the order: a SIMD spends ~4 cycles per issued instruction whichever
// wave it comes from, so what overlap can win is only the matrix pipe's own 32-cycle occupancy.
// NSEL (FAST; option "flash_nsel", default on since round 4): staging without
// per-element selects.  The default lstore zeroes every chunk that is not real data with v_cndmask (4 per 16-byte chunk, K and V: 32 VALU per
// thread per tile at d = 128, in a loop that is VALU-issue bound).  Here chunks that fetch nothing (the padding chunk of d = 40 / 80 rows) point their
// buffer offset beyond num_records, so the load itself returns zeros, and full tiles (every key < Lk; a wave-uniform test) are stored as loaded;
// only the ragged last tile takes the select path.
// PK (round 6; option "flash_pk"; launches whose kernel subtracts the running max / adds the row sums on the VALU, i.e. every head dim but the d = 40 max-slot kernel):
// those two passes over the 32 scores of a lane as PACKED f32 operations (v_pk_add_f32: two scores per issued instruction) — 33 v_sub + 34 v_add per
// tile become 16 + 17 in a loop that is bound by instruction issue (d = 64: 165 non-MFMA VALU instructions per 64-key tile next to 16 MFMAs before, 133 with it).
// The row sum is accumulated as two interleaved partial sums (even / odd score registers): the same terms in another association than the serial chain.
// Measured SLOWER (profiles/r09a_flash_pk_rejected.txt: d = 64 138 -> 144 us, d = 128 329 -> 345 us): a v_pk_add_f32 is not cheaper than the two v_add_f32 it replaces here.
// SM = softmax arithmetic variant bits: 1 = PK; 2 = MINIT (one query block per wave): the QK^T accumulator is INITIALISED with -m_run (sixteen registers holding the
// lane's query's -m, rewritten only when the running max moves), so the scores leave the MFMA relative to the max — the d = 40 max slot's effect without a k-slot — and the
// 32 v_sub per tile disappear; 4 = DOT2: the row sum from the f16-rounded P (the values the numerator uses) with v_dot2_f32_f16 against (1, 1): 16 instead of 32 additions.
template <int DKP, int NDV, bool FAST, int ABL = 0, bool MSLOT = false, int QB = 1, bool VPF = false, bool VTR = false, bool OVL = false, bool NSEL = false, int SM = 0>
__global__ __launch_bounds__(256, QB == 2 ? (DKP <= 96 ? 2 : 1) : (DKP <= 64 ? ((VPF && !(SM & 2)) ? 2 : FA_OCC_SMALL) : (DKP <= 128 ? 2 : 1))) void k_flash_attn(FAArgs g) {
    static_assert(QB == 1 || FAST, "two query blocks per wave: FAST staging only");
    static_assert(!MSLOT || DKP == 48 || DKP == 80, "max slot: d = 40 on the 48-wide tile, d = 64 on the 80-wide tile");
    constexpr int MS_HI = DKP == 48 ? 1 : 0;  // lane half holding element d = D of the last k-step (the max slot)
    static_assert(!OVL || (QB == 2 && DKP == 48 && NDV == 2 && VPF && FAST && ABL == 0), "overlapped issue order: the two-block d <= 48 kernel only");
    static_assert(!NSEL || FAST, "select-free staging: FAST staging only");
    constexpr bool PK = (SM & 1) != 0, MINIT = (SM & 2) != 0, DOT2 = (SM & 4) != 0;
    static_assert(!(MINIT || DOT2) || (!OVL && !MSLOT && ABL == 0), "MINIT / DOT2: the phase-by-phase kernels without the max slot");
    static_assert(!MINIT || (QB == 1 && DKP / 16 <= 6), "MINIT: sixteen more registers per query block — one block per wave only; written for the whole-tile fragment path (d <= 96)");
    constexpr bool REL = MSLOT || MINIT;  // scores leave the MFMA relative to the running max
    static_assert(!VTR || FAST, "row-major V tiles: FAST staging only");
    constexpr int VRS  = fa_vtr_stride(NDV);             // VTR: V tile row stride (halfs)
    constexpr int VT_H = VTR ? FA_KT * VRS : NDV * 32 * FA_VTS;  // halfs of one V tile
    constexpr int QW   = 32 * QB;                        // queries per wave
    constexpr int QWG  = 4 * QW;                         // queries per workgroup
    constexpr int KS   = DKP / 16;                       // MFMA k-steps over the head dim
    constexpr int KROW = DKP + 8;                        // K tile row stride (halfs)
    constexpr int DCH  = DKP / 8;                        // 8-wide d chunks per key
    constexpr int NCH  = FAST ? (FA_KT * DCH + 255) / 256 : 1;  // prefetched chunks per thread (K and V each)
    // FAST: two K/V tile buffers — tile t+1 is written while tile t is read, ONE barrier per tile (waves were parked at the two
    // barriers of the single-buffer loop 47 % of their cycles, profiles/r01g_pmc_flash.txt)
    constexpr int TILE_H = FA_KT * KROW + VT_H;  // halfs per tile buffer
    constexpr int NBUF   = FAST ? 2 : 1;
    __shared__ __attribute__((aligned(16))) _Float16 smem[NBUF * TILE_H];
    _Float16* Ks = smem;
    _Float16* Vt = smem + FA_KT * KROW;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi   = lane >> 5;
    // workgroup id -> (head-image hn, query block): consecutive ids go round-robin over the 8 XCDs, so with the plain (query block, hn) grid the
    // heads of one token row land on different L2s at different times — the head-interleaved f16 output (80-byte pieces of a 640-byte
    // token row at d = 40) is then written as 8 partial lines from up to 8 L2s.  grp > 0: the grp heads of one (image, query block) unit run
    // back to back on ONE XCD (id = (unit / 8 * grp + head) * 8 + unit % 8), so an output line is completed inside one L2 and a K/V
    // stream is shared by 3x more of the XCD's resident workgroups.
    int hn, qb;
    if (g.grp > 0) {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int h = j % g.grp, unit = (j / g.grp) * 8 + xcd;
        if (unit >= g.units) return;
        const int nrb = (g.Lq + QWG - 1) / QWG;
        hn = (unit / nrb) * g.grp + h;
        qb = unit % nrb;
    } else {
        hn = blockIdx.y;
        qb = blockIdx.x;
    }
    const int q0   = qb * QWG + wave * QW;              // first query of this wave; query block b starts at q0 + 32 * b
    const int qi   = q0 + (lane & 31);

    // ---- Q fragments (B operand of S^T): lane holds Q[qi][ks*16 + hi*8 .. +8] as f16.
    // q_vec (rows d-contiguous f32, 16-byte aligned, D % 4 == 0 — the head-major projection output): the workgroup's 128 rows are fetched with
    // coalesced float4 loads and handed to the lanes through LDS (the region aliases the K/V tiles, which are staged afterwards).  Per-lane
    // row reads (32 rows x 2 halves per instruction, 4 useful bytes of every 64-byte segment) cost ~1500 TA cycles per wave: at Lk = 77
    // (cross-attention, 2 tiles per workgroup) that was most of the kernel.
    half8_t qf[QB][KS];
    if (g.q_f16) {  // f16 head-major Q written by the projection GEMM for this launch (rows d-contiguous, 16-byte aligned, D % 8 == 0)
        constexpr int QROW = DKP + 4, C8 = DKP / 8;
        static_assert(QWG * QROW <= NBUF * TILE_H, "Q staging must fit the K/V tile buffers");
        const char* qblk = g.q + (int64_t)hn * g.q_nb2 + (int64_t)(qb * QWG) * g.q_nb1;
        for (int e = threadIdx.x; e < QWG * C8; e += 256) {
            const int row = e / C8, c8 = e - row * C8;
            half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qb * QWG + row < g.Lq && c8 * 8 < g.D) v = *(const half8_t*)(qblk + (int64_t)row * g.q_nb1 + c8 * 16);
            half4_t a, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = (_Float16)((float)v[j] * g.scale_log2e);
                b[j] = (_Float16)((float)v[4 + j] * g.scale_log2e);
            }
            *(half4_t*)&smem[row * QROW + c8 * 8]     = a;
            *(half4_t*)&smem[row * QROW + c8 * 8 + 4] = b;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const _Float16* p = &smem[(wave * QW + b * 32 + (lane & 31)) * QROW + ks * 16 + hi * 8];
                const half4_t a = *(const half4_t*)p, c = *(const half4_t*)(p + 4);
                qf[b][ks] = (half8_t){a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            }
        __syncthreads();
    } else if (g.q_vec) {
        constexpr int QROW = DKP + 4, C4 = DKP / 4;  // +4 halfs: 8-byte aligned rows, 2-way conflicts at worst on the one-time fragment reads
        static_assert(QWG * QROW <= NBUF * TILE_H, "Q staging must fit the K/V tile buffers");
        const char* qblk = g.q + (int64_t)hn * g.q_nb2 + (int64_t)(qb * QWG) * g.q_nb1;
        for (int e = threadIdx.x; e < QWG * C4; e += 256) {
            const int row = e / C4, c4 = e - row * C4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qb * QWG + row < g.Lq && c4 * 4 < g.D) v = *(const float4*)(qblk + (int64_t)row * g.q_nb1 + c4 * 16);
            half4_t h;
            h[0] = (_Float16)(v.x * g.scale_log2e);  // scores come out of the MFMA in log2 units
            h[1] = (_Float16)(v.y * g.scale_log2e);
            h[2] = (_Float16)(v.z * g.scale_log2e);
            h[3] = (_Float16)(v.w * g.scale_log2e);
            *(half4_t*)&smem[row * QROW + c4 * 4] = h;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const _Float16* p = &smem[(wave * QW + b * 32 + (lane & 31)) * QROW + ks * 16 + hi * 8];
                const half4_t a = *(const half4_t*)p, c = *(const half4_t*)(p + 4);
                qf[b][ks] = (half8_t){a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            }
        __syncthreads();
    } else {
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const int qib     = qi + 32 * b;
            const float* qrow = (const float*)(g.q + (int64_t)min(qib, g.Lq - 1) * g.q_nb1 + (int64_t)hn * g.q_nb2);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int d  = ks * 16 + hi * 8 + j;
                    qf[b][ks][j] = (_Float16)((d < g.D && qib < g.Lq) ? qrow[d] * g.scale_log2e : 0.f);
                }
            }
        }
    }

    float16_t o[QB][NDV];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int nb = 0; nb < NDV; ++nb) o[b][nb] = (float16_t){0};
        m_run[b] = REL ? 0.f : -INFINITY;  // MSLOT: Q's max slot starts at 0 and the first tile always moves the max
        l_run[b] = 0.f;
    }

    float16_t negm[MINIT ? QB : 1];  // MINIT: -m_run of this lane's query in all sixteen registers (the C operand of the first QK^T MFMA of each key block)
#pragma unroll
    for (int b = 0; b < (MINIT ? QB : 1); ++b) negm[b] = (float16_t){0};

    const char* kbase = g.k + (int64_t)hn * g.k_nb2;
    const char* vbase = g.v + (int64_t)hn * g.v_nb2;

    // ---- FAST: split staging (T14): registers hold the next tile's K and V chunks (8 halfs = 16 B each)
    half8_t kreg[NCH], vreg[NCH];
    const int nd8 = g.D / 8;  // valid 8-wide chunks per row
    // Row sums for free: when the head dim leaves a padded output column (DV < NDV*32; d = 40, 80: yes, d = 64, 128, 160: no), row DV
    // of V^T is set to ones and column DV of the PV accumulator becomes sum_k P[q][k] — in the same f16-rounded P the numerator
    // uses — instead of 32 VALU adds per lane per tile.  (FAST staging needs D % 8 == 0 for the row to sit at a chunk start.)
    // (OVL: the FAST d <= 48 launch always has the column — D = DV, D % 8 == 0 and D < 64 are dispatch conditions — and a compile-time constant
    // keeps the tile body free of the branch that would cut its scheduling regions)
    const bool has_ones     = OVL ? true : (g.DV < NDV * 32 && g.D == g.DV && (!FAST || g.D % 8 == 0));
    const bool ones_in_tile = has_ones && g.DV < DKP;  // the row is (re)written by the per-tile staging; else set once below
    // K chunks: thread e -> (key = e / DCH, chunk = e % DCH): row-major 16-byte LDS writes.  V chunks: (key = e % 64, chunk = e / 64):
    // lanes run along keys so the 8 transposing 2-byte LDS writes of a chunk are bank-contiguous (the (e / DCH, e % DCH) mapping
    // put a wave's 64 lanes on ~12 banks: 31 % of all LDS cycles were conflict cycles).
    // per-thread chunk coordinates are tile-invariant: 32-bit byte offsets against a wave-uniform tile base (SGPR base + VGPR offset loads)
    uint32_t koff[NCH], voff[NCH];  // always a readable address: lanes / chunks that fetch nothing point at the tile's first bytes and are zeroed by lstore
    bool kone[NCH];             // MSLOT: this chunk starts at d = D (the max slot)
    int kkey[NCH], vkey_[NCH];  // key index inside the tile, or FA_KT (never valid) for chunks this thread does not fetch
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e   = threadIdx.x + c * 256;
        const int key = e / DCH, ch = e - key * DCH;
        const int vkey = VTR ? key : (e & (FA_KT - 1)), vch = VTR ? ch : (e >> 6);
        kkey[c]  = (e < FA_KT * DCH && ch < nd8) ? key : FA_KT;
        kone[c]  = e < FA_KT * DCH && ch == nd8;
        vkey_[c] = (e < FA_KT * DCH && vch < nd8) ? vkey : FA_KT;
        koff[c]  = kkey[c] < FA_KT ? (uint32_t)key * (uint32_t)g.k_nb1 + (uint32_t)ch * 16u : (NSEL ? 0x7fffff00u : 0u);  // NSEL: out of range -> zeros
        voff[c]  = vkey_[c] < FA_KT ? (uint32_t)vkey * (uint32_t)g.v_nb1 + (uint32_t)vch * 16u : (NSEL ? 0x7fffff00u : 0u);
    }
    // registers <- global.  BUFFER loads: uniform descriptor (one head's K / V rows, num_records = Lk rows: keys beyond Lk read as zeros for
    // free), per-lane 32-bit offsets that never change, the tile offset in an SGPR — no per-tile address arithmetic and NOTHING touches the
    // destination registers before lstore.  History (disassembly, round 3): with "zero, then flat load where valid" every load was preceded by an
    // s_waitcnt vmcnt(0) (the zeroing v_mov / the 64-bit address computed INTO the destination registers needs their previous load retired, and
    // the counter cannot tell loads apart), so the 2 NCH loads of a tile paid their latencies one after the other, and inserting the constant
    // 1s right behind the loads added another wait each — the register prefetch hid nothing.
    // (the descriptors' words are wave-uniform by construction; said explicitly, because under SGPR pressure — the two-block d = 64 instantiation — the compiler kept them in
    // VGPRs and wrapped every tile's loads in waterfall loops)
    auto uni_ptr = [](const char* p) {
        const uint64_t u = (uint64_t)(uintptr_t)p;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi_ << 32) | lo);
    };
    kbase = uni_ptr(kbase);
    vbase = uni_ptr(vbase);
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)min((int64_t)g.Lk * g.k_nb1, (int64_t)0x7fffffff), 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)min((int64_t)g.Lk * g.v_nb1, (int64_t)0x7fffffff), 0x00020000);
    auto gload = [&](int kt) {
        const int sk = kt * (int)g.k_nb1, sv = kt * (int)g.v_nb1;  // wave-uniform tile offsets
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            kreg[c] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rsK, (int)koff[c], sk, 0));
            vreg[c] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rsV, (int)voff[c], sv, 0));
        }
    };
    // LDS <- registers (tile starting at key kt): zero what is not real data, insert the constant 1s of the max slot / ones row
    auto lstore = [&](int buf, int kt) {
        _Float16* ks = Ks + buf * TILE_H;
        _Float16* vt = Vt + buf * TILE_H;
        const int left = g.Lk - kt;
        if constexpr (NSEL) {
            if (left >= FA_KT) {  // wave-uniform: a full tile is stored as loaded (padding chunks were read as zeros by the range check)
                asm volatile("" ::: "memory");  // keeps this a branch: if-converted, the two paths merge back into per-element selects
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int e = threadIdx.x + c * 256;
                    if (e < FA_KT * DCH) {
                        const int key = e / DCH, ch = e - key * DCH;
                        half8_t kv = kreg[c], vv = vreg[c];
                        if (MSLOT && kone[c]) kv[0] = (_Float16)1.0f;
                        if (ones_in_tile && ch == nd8) vv[0] = (_Float16)1.0f;
                        *(half8_t*)&ks[key * KROW + ch * 8] = kv;
                        static_assert(!NSEL || VTR, "select-free staging is written for the row-major V tile");
                        *(half8_t*)&vt[key * VRS + ch * 8] = vv;
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e = threadIdx.x + c * 256;
            if (e < FA_KT * DCH) {
                const int key = e / DCH, ch = e - key * DCH;
                const int vkey = VTR ? key : (e & (FA_KT - 1)), vch = VTR ? ch : (e >> 6);
                const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                half8_t kv = kkey[c] < left ? kreg[c] : z, vv = vkey_[c] < left ? vreg[c] : z;
                if (MSLOT && kone[c]) kv[0] = (_Float16)1.0f;                   // K[key][D] = 1 (keys beyond Lk are masked after the MFMA)
                if (ones_in_tile && vch == nd8) vv[0] = (_Float16)1.0f;         // V^T row DV = 1: PV accumulates the row sums
                *(half8_t*)&ks[key * KROW + ch * 8] = kv;
                if constexpr (VTR) {
                    *(half8_t*)&vt[vkey * VRS + vch * 8] = vv;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) vt[(vch * 8 + j) * FA_VTS + vkey] = vv[j];
                }
            }
        }
    };
    // ---- generic staging (any strides, f16 or f32 sources): straight to LDS
    auto stage_generic = [&](int kt) {
        for (int e = threadIdx.x; e < FA_KT * DCH; e += 256) {
            const int key = e / DCH, ch = e - key * DCH;
            const int kk  = kt + key;
            half8_t h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = ch * 8 + j;
                float v     = 0.f;
                if (kk < g.Lk && d < g.D) {
                    const char* p = kbase + (int64_t)kk * g.k_nb1;
                    v             = g.kv_f16 ? __half2float(((const __half*)p)[d]) : ((const float*)p)[d];
                }
                h[j] = (_Float16)v;
            }
            *(half8_t*)&Ks[key * KROW + ch * 8] = h;
        }
        // V: lanes along keys (coalesced for the key-contiguous vT of the manual chain, conflict-free 2-byte LDS writes)
        for (int e = threadIdx.x; e < FA_KT * DCH; e += 256) {
            const int key = e & (FA_KT - 1), ch = e >> 6;
            const int kk  = kt + key;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = ch * 8 + j;
                float v     = 0.f;
                if (kk < g.Lk && d < g.DV) {
                    const char* p = vbase + (int64_t)kk * g.v_nb1 + (int64_t)d * g.v_nb0;
                    v             = g.kv_f16 ? __half2float(*(const __half*)p) : *(const float*)p;
                }
                if (ones_in_tile && d == g.DV) v = 1.0f;
                Vt[d * FA_VTS + key] = (_Float16)v;
            }
        }
    };

    // padded V^T rows (d >= DKP) are never staged: clear them once so the masked output columns stay finite
#pragma unroll
    for (int b = 0; b < NBUF; ++b)
        for (int e = threadIdx.x; e < VT_H / 2; e += 256) ((uint32_t*)(Vt + b * TILE_H))[e] = 0u;
    if (has_ones && !ones_in_tile) {
        __syncthreads();
        if (threadIdx.x < FA_KT * NBUF) Vt[(threadIdx.x >> 6) * TILE_H + (VTR ? (threadIdx.x & 63) * VRS + g.DV : g.DV * FA_VTS + (threadIdx.x & 63))] = (_Float16)1.0f;
    }
    if (FAST) {
        gload(0);
        __syncthreads();  // the clears above
        lstore(0, 0);
        if (FA_KT < g.Lk) gload(FA_KT);
        __syncthreads();
    }
    int buf = 0;
    // one 64-key tile.  TAIL (keys beyond Lk masked) is a compile-time flag: the hot loop over full tiles carries no mask code and no branch that
    // would cut its basic blocks; a ragged last tile runs the second instantiation once.
    auto tile = [&](const int kt, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if (FAST) {
            // tile kt sits in buffer buf (visible since the barrier that ended the previous iteration); the registers hold tile
            // kt+64: park it in the other buffer now, then fetch kt+128 — both overlap this tile's MFMAs
            if (ABL != 2) {
                if (kt + FA_KT < g.Lk) lstore(buf ^ 1, kt + FA_KT);
                if (kt + 2 * FA_KT < g.Lk) gload(kt + 2 * FA_KT);
            }
        } else {
            __syncthreads();  // every wave finished reading the previous tile
            stage_generic(kt);
            __syncthreads();
        }
        const _Float16* Kc = Ks + (FAST ? buf * TILE_H : 0);
        const _Float16* Vc = Vt + (FAST ? buf * TILE_H : 0);

        // ---- per query block: S^T = K Q^T (rows i = key, cols j = query) -> online softmax -> P packed to f16.  Block b+1's QK^T MFMAs are
        // independent of block b's softmax VALU work, so the two streams overlap; only ONE block's 32 score registers are live at a time.
        half8_t pa[QB][4];
        half4_t vq0[4], vq1[4];  // VPF: ring of four V fragments (two 8-byte halves each)
        // B fragment (16 keys of k-step t, output columns nb * 32 + (lane & 31)) as two 4-key halves: keys t*16 + 4 hi .. and the same 8 further on
        const int vtr_lane = (4 * hi + ((lane & 15) >> 2)) * VRS + ((lane >> 4) & 1) * 16 + 4 * (lane & 3);
        auto vread = [&](int t, int nb, half4_t& a, half4_t& c) {
            if constexpr (VTR) {
                const _Float16* p = Vc + vtr_lane + t * 16 * VRS + nb * 32;
                a = lds_read_tr16(p);
                c = lds_read_tr16(p + 8 * VRS);
            } else {
                const _Float16* vrow = &Vc[(nb * 32 + (lane & 31)) * FA_VTS + t * 16 + 4 * hi];
                a = *(const half4_t*)vrow;
                c = *(const half4_t*)(vrow + 8);
            }
        };
        constexpr bool SHARE_KF = QB == 2 && KS <= 4;
        half8_t kf[2][KS <= 6 ? KS : 1];
        // ---- online softmax for query (lane & 31) of each block; this lane holds keys kb*32 + (r&3)+8*(r>>2)+4*hi.  The loop is VALU-bound at
        // d = 40 (32 exps per lane per tile and block vs 14 MFMAs), so every instruction counts: scores arrive pre-scaled (Q carries
        // scale*log2e), exp is the bare v_exp_f32, and the running max is DEFERRED — it only moves when a query's tile max exceeds
        // it by more than FA_THR (2^8: P <= 256 stays exact enough in f16 and far from its range), which makes the accumulator
        // rescale (16 cross-lane fetches + 16*NDV multiplies) rare instead of per tile.  The vote uses each lane's OWN 32 keys (a row's max
        // exceeds the bar iff one of its two halves does): the cross-half exchange happens only on the rare path.
        if constexpr (OVL) {
            float16_t s0[2], s1[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const half8_t*)&Kc[(kb * 32 + (lane & 31)) * KROW + ks * 16 + hi * 8];
            __builtin_amdgcn_sched_barrier(0);
            auto qk = [&](const int b, float16_t (&sc)[2]) {
                sc[0] = (float16_t){0};
                sc[1] = (float16_t){0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[b][ks], sc[kb], 0, 0, 0);
            };
            auto tailmask = [&](float16_t (&sc)[2]) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= g.Lk) sc[kb][r] = -INFINITY;
            };
            // tile max and the (rare) move of the running max — the code of the default path, on one block's scores
            auto maxrare = [&](const int b, float16_t (&sc)[2]) {
                float tmax = sc[0][0];
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, sc[0][r]), sc[1][r]);
                if (MSLOT ? (kt == 0 || __any(tmax > FA_THR)) : __any(tmax > m_run[b] + FA_THR)) {
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                    float alpha;
                    if constexpr (MSLOT) {
                        const float m_new = fminf((float)(_Float16)fminf(m_run[b] + (kt == 0 ? tmax : fmaxf(tmax, 0.f)), 65504.f), 65504.f);
                        const float delta = m_new - m_run[b];
                        alpha             = __builtin_amdgcn_exp2f(-delta);
                        m_run[b]          = m_new;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
                        if (hi) qf[b][KS - 1][0] = (_Float16)(-m_new);
                    } else {
                        const float m_new = fmaxf(m_run[b], tmax);
                        alpha             = __builtin_amdgcn_exp2f(m_run[b] - m_new);
                        m_run[b]          = m_new;
                    }
                    l_run[b] *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row  = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const float ar = __shfl(alpha, row, 64);
#pragma unroll
                        for (int nb = 0; nb < NDV; ++nb) o[b][nb][r] *= ar;
                    }
                }
            };
            auto expcvt = [&](const int b, float16_t (&sc)[2]) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kb][r] = __builtin_amdgcn_exp2f(MSLOT ? sc[kb][r] : sc[kb][r] - m_run[b]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kb = t >> 1, rb = (t & 1) * 8;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const half2_t h2 = __builtin_convertvector((float2_t){sc[kb][rb + j], sc[kb][rb + j + 1]}, half2_t);
                        pa[b][t][j]     = h2[0];
                        pa[b][t][j + 1] = h2[1];
                    }
                }
            };
            qk(0, s0);
            if (TAIL) tailmask(s0);
            maxrare(0, s0);
            // ---- region A: QK^T of block 1, the exponentials / f16 packing of block 0 in its gaps
            qk(1, s1);
            expcvt(0, s0);
            // pin block 0's P here: left alone, the compiler sinks the (side-effect-free) exponentials down to their first user, the P V region
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(pa[0][t]));
#pragma unroll
            for (int i = 0; i < 2 * KS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            if (TAIL) tailmask(s1);
#pragma unroll
            for (int i = 0; i < 4; ++i) vread(i / NDV, i % NDV, vq0[i], vq1[i]);  // in flight during block 1's max chain
            __builtin_amdgcn_sched_barrier(0);
            maxrare(1, s1);
            // ---- regions B + C: P V of block 0 with block 1's exponentials in its gaps, then P V of block 1 (the fragment ring runs through both)
            expcvt(1, s1);
            constexpr int NFR = 4 * NDV;
#pragma unroll
            for (int k = 0; k < 2 * NFR; ++k) {
                const int b = k / NFR, i = k % NFR, t = i / NDV, nb = i % NDV;
                const half4_t v0 = vq0[k & 3], v1 = vq1[k & 3];
                const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[b][t], vf, o[b][nb], 0, 0, 0);
                if (k + 4 < 2 * NFR) vread(((k + 4) % NFR) / NDV, ((k + 4) % NFR) % NDV, vq0[k & 3], vq1[k & 3]);
            }
#pragma unroll
            for (int k = 0; k < 2 * NFR; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (k < NFR) {
                    __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                if (k + 4 < 2 * NFR) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else {
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float16_t s[2];
            if constexpr (KS <= 6) {
                // all K fragments of the tile are requested before the first MFMA (left to itself the compiler issued read -> wait -> MFMA
                // one at a time through a single register set: the LDS latency sat in front of every MFMA).  SHARE_KF (two query blocks,
                // d <= 48): the fragments stay in registers for the second block — no second read, and its MFMAs start without an LDS wait
                if (!SHARE_KF || b == 0) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const half8_t*)&Kc[(kb * 32 + (lane & 31)) * KROW + ks * 16 + hi * 8];
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (MINIT) {
                    // D = A B + C with C = the -m_run registers and D = the score registers: written as the instruction itself, because the builtin ties D to C and
                    // the compiler then copies the sixteen registers first (24 moves per tile: most of what the 32 subtractions cost)
                    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s[0]) : "v"(kf[0][0]), "v"(qf[b][0]), "v"(negm[b]));
                    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s[1]) : "v"(kf[1][0]), "v"(qf[b][0]), "v"(negm[b]));
                } else {
                    s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][0], qf[b][0], (float16_t){0}, 0, 0, 0);
                    s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1][0], qf[b][0], (float16_t){0}, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 1; ks < KS; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[b][ks], s[kb], 0, 0, 0);
            } else if constexpr (VPF && KS == 8) {
                // d = 128: groups of four fragments, the next group's reads in flight during the current group's MFMAs
                constexpr int NG = 2 * KS / 4;  // groups over (kb, ks)
                half8_t ka[4], kbq[4];
                auto rdg = [&](half8_t (&dst)[4], int gi) {
                    const int kb = (gi * 4) / KS, ks0 = (gi * 4) % KS;
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j] = *(const half8_t*)&Kc[(kb * 32 + (lane & 31)) * KROW + (ks0 + j) * 16 + hi * 8];
                };
                s[0] = (float16_t){0};
                s[1] = (float16_t){0};
                rdg(ka, 0);
#pragma unroll
                for (int gi = 0; gi < NG; gi += 2) {
                    rdg(kbq, gi + 1);
                    {
                        const int kb = (gi * 4) / KS, ks0 = (gi * 4) % KS;
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[j], qf[b][ks0 + j], s[kb], 0, 0, 0);
                    }
                    if (gi + 2 < NG) rdg(ka, gi + 2);
                    {
                        const int kb = ((gi + 1) * 4) / KS, ks0 = ((gi + 1) * 4) % KS;
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kbq[j], qf[b][ks0 + j], s[kb], 0, 0, 0);
                    }
                }
                static_assert(NG == 4, "the issue pattern below is written for KS = 8");
                // issue order: the reads of groups 0 and 1, then MFMAs of group g followed by the reads of group g + 2
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            } else {  // d = 128, 160: the fragments of a whole tile do not fit next to the accumulators
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    s[kb] = (float16_t){0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const half8_t kf = *(const half8_t*)&Kc[(kb * 32 + (lane & 31)) * KROW + ks * 16 + hi * 8];
                        s[kb]            = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[b][ks], s[kb], 0, 0, 0);
                    }
                }
            }
            if (TAIL) {  // tail tile: mask keys >= Lk
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= g.Lk) s[kb][r] = -INFINITY;
            }
            if (VPF && b == QB - 1) {  // the first four V fragments of this tile: in flight during the (last block's) softmax
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vread(i / NDV, i % NDV, vq0[i], vq1[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float tmax = s[0][0];
            if (ABL != 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, s[0][r]), s[1][r]);  // v_max3_f32
            }
            if (REL ? (kt == 0 || __any(tmax > FA_THR)) : (ABL != 1 && __any(tmax > m_run[b] + FA_THR))) {
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                float alpha;
                if constexpr (MINIT) {
                    // as the max slot below, without its f16 constraint: -m enters the MFMA as an f32 accumulator value
                    const float m_new = m_run[b] + (kt == 0 ? tmax : fmaxf(tmax, 0.f));
                    const float delta = m_new - m_run[b];
                    alpha             = kt == 0 ? 0.f : __builtin_amdgcn_exp2f(-delta);  // (first tile: O and l are zero)
                    m_run[b]          = m_new;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
#pragma unroll
                    for (int r = 0; r < 16; ++r) negm[MINIT ? b : 0][r] = -m_new;
                } else if constexpr (MSLOT) {
                    // the scores are relative to m_run already: move the max by delta (rounded so that the new max is an f16 value), re-base this
                    // tile.  The new max is clamped to the finite f16 range: the slot holds -m as f16, and +-inf there would turn every later
                    // score of the row into NaN; any consistent offset is a valid softmax shift
                    const float m_new = fminf((float)(_Float16)fminf(m_run[b] + (kt == 0 ? tmax : fmaxf(tmax, 0.f)), 65504.f), 65504.f);
                    const float delta = m_new - m_run[b];
                    alpha             = __builtin_amdgcn_exp2f(-delta);
                    m_run[b]          = m_new;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
                    if (hi == MS_HI) qf[b][KS - 1][0] = (_Float16)(-m_new);  // d = 40: k-step 2, upper lane half, element 0; d = 64 on the 80-wide tile: k-step 4, lower half
                } else {
                    const float m_new = fmaxf(m_run[b], tmax);
                    alpha             = __builtin_amdgcn_exp2f(m_run[b] - m_new);  // m_run = -inf on the first tile -> 0
                    m_run[b]          = m_new;
                }
                l_run[b] *= alpha;
                // rescale O rows: row i of the accumulator belongs to query lane i -> fetch its alpha
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row  = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float ar = __shfl(alpha, row, 64);
#pragma unroll
                    for (int nb = 0; nb < NDV; ++nb) o[b][nb][r] *= ar;
                }
            }
            if (ABL != 1) {
                if constexpr (PK && !REL) {
                    const float2_t mm = {m_run[b], m_run[b]};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {  // v_pk_add_f32 (neg): two scores per instruction
                            const float2_t d2 = (float2_t){s[kb][r], s[kb][r + 1]} - mm;
                            s[kb][r]          = __builtin_amdgcn_exp2f(d2[0]);
                            s[kb][r + 1]      = __builtin_amdgcn_exp2f(d2[1]);
                        }
                } else {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(REL ? s[kb][r] : s[kb][r] - m_run[b]);
                }
            }
            if (!has_ones && !DOT2) {
                if constexpr (PK) {
                    // two independent chains (a dependent v_pk_add_f32 needs a wait state after its producer: a single chain is issued with an s_nop per link)
                    float2_t pa2 = {s[0][0], s[0][1]}, pb2 = {s[1][0], s[1][1]};
#pragma unroll
                    for (int r = 2; r < 16; r += 2) {
                        pa2 += (float2_t){s[0][r], s[0][r + 1]};
                        pb2 += (float2_t){s[1][r], s[1][r + 1]};
                    }
                    pa2 += pb2;
                    l_run[b] += pa2[0] + pa2[1];
                } else {
                    float psum = 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) psum += s[kb][r];
                    l_run[b] += psum;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {  // P in f16, in the accumulator's own key order: the A operand of P V with no data movement
                const int kb = t >> 1, rb = (t & 1) * 8;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {  // v_cvt_pk_f16_f32
                    const half2_t h2 = __builtin_convertvector((float2_t){s[kb][rb + j], s[kb][rb + j + 1]}, half2_t);
                    pa[b][t][j]     = h2[0];
                    pa[b][t][j + 1] = h2[1];
                }
            }
            if constexpr (DOT2) {
                if (!has_ones) {  // four independent chains of v_dot2_f32_f16 over the packed P
                    const half2_t one2 = {(_Float16)1.0f, (_Float16)1.0f};
                    float ps4[4]       = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < 8; j += 2) ps4[j >> 1] = __builtin_amdgcn_fdot2((half2_t){pa[b][t][j], pa[b][t][j + 1]}, one2, ps4[j >> 1], false);
                    l_run[b] += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
                }
            }
        }
        // ---- O += P V : 4 k-steps of 16 keys; k-slot order = accumulator key order (see header); every V fragment feeds the QB blocks
        if constexpr (VPF) {
            constexpr int NFR = 4 * NDV;
#pragma unroll
            for (int i = 0; i < NFR; ++i) {
                const int t = i / NDV, nb = i % NDV;
                const half4_t v0 = vq0[i & 3], v1 = vq1[i & 3];
                const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int b = 0; b < QB; ++b) o[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[b][t], vf, o[b][nb], 0, 0, 0);
                if (i + 4 < NFR) vread((i + 4) / NDV, (i + 4) % NDV, vq0[i & 3], vq1[i & 3]);
            }
#pragma unroll
            for (int i = 0; i < NFR; ++i) {  // the fragment's MFMA(s), then the two reads that refill its ring slot
                __builtin_amdgcn_sched_group_barrier(0x008, QB, 0);
                if (i + 4 < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int nb = 0; nb < NDV; ++nb) {
                half4_t v0, v1;
                vread(t, nb, v0, v1);
                half8_t vf;
                vf[0] = v0[0];
                vf[1] = v0[1];
                vf[2] = v0[2];
                vf[3] = v0[3];
                vf[4] = v1[0];
                vf[5] = v1[1];
                vf[6] = v1[2];
                vf[7] = v1[3];
#pragma unroll
                for (int b = 0; b < QB; ++b) o[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[b][t], vf, o[b][nb], 0, 0, 0);
            }
        }
        }  // !OVL
        if (FAST) {
            __syncthreads();  // tile kt+64 is complete in the other buffer; everybody is done reading this one
            if (ABL != 2) buf ^= 1;
        }
    };
    {
        int kt = 0;
        for (; kt + FA_KT <= g.Lk; kt += FA_KT) tile(kt, std::false_type{});
        if (kt < g.Lk) tile(kt, std::true_type{});
    }

    // ---- finalise: divide by the row sum (both lane halves), write [d] contiguous
    const int nb_l = g.DV >> 5, lane_l = (g.DV & 31) + 32 * hi;  // has_ones: accumulator column DV holds the row sums
    const int hh = g.H > 0 ? hn % g.H : hn, nn = g.H > 0 ? hn / g.H : 0;  // wave-uniform
    char* obase          = g.dst ? (char*)g.dst + (int64_t)hh * g.dst_nb_h + (int64_t)nn * g.dst_nb_n : nullptr;
    _Float16* obase16    = g.dst16 ? g.dst16 + (int64_t)nn * g.Lq * g.ld16 + (int64_t)hh * g.DV : nullptr;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        const float inv   = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row  = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float ir = __shfl(inv, row, 64);
            if (has_ones) {
                float lsum = 0.f;
#pragma unroll
                for (int nb = 0; nb < NDV; ++nb)
                    if (nb == nb_l) lsum = o[b][nb][r];
                lsum = __shfl(lsum, lane_l, 64);
                ir   = lsum > 0.f ? 1.0f / lsum : 0.f;
            }
            const int q = q0 + 32 * b + row;
            if (q >= g.Lq) continue;
#pragma unroll
            for (int nb = 0; nb < NDV; ++nb) {
                const int d = nb * 32 + (lane & 31);
                if (d >= g.DV) continue;
                const float val = o[b][nb][r] * ir;
                if (obase) *(float*)(obase + (int64_t)q * g.dst_nb_q + d * 4) = val;
                if (obase16) obase16[(int64_t)q * g.ld16 + d] = (_Float16)val;
            }
        }
    }
}

// =====================================================================================================================================
// k_flash_short — attention onto a SHORT key sequence (64 < Lk <= 96: the 77 text tokens every SD1.x / SDXL cross-attention reads), d <= 64.
// Written at the end of round 3, first run in round 4 (scripts/flash_check.py short; index logic replayed lane by lane in tests/test_kernel_logic.py):
// SD1.5 64x64-level cross-attention 81 -> 40 us, SDXL's 20.2 -> 16.6 us, SD1.5 step -1 % (profiles/r05a_flash_short.txt).  Its first GPU run
// returned garbage: the output stores sat behind a lambda taking the accumulator register index as an argument, and hipcc (ROCm 7.2) stored
// element 0 sixteen times — the stores are plain unrolled loops now.
// Why: the tile kernel spends a cross-attention launch on fixed costs — per 128 queries one workgroup stages Q through LDS, stages two K / V tiles
// (the second holds 13 valid keys) and passes five barriers for 28 MFMAs: 80 us per SD1.5 launch at the 64x64 level (16 launches per step) for
// 126 MB of traffic, i.e. 1.6 TB/s.  Here the whole K and V of one head live in REGISTERS: a workgroup stages the keys once (K row-major and
// already multiplied by scale * log2(e), V row-major for the transposing read), every wave pulls all its MFMA fragments (3 key blocks x KS K
// fragments, 6 k-steps x 2 V fragments) and then walks g.qi blocks of 32 queries with no LDS access and no barrier.  The block loop is written
// for instruction count (these kernels are issue-bound): Q fragments are 16-byte buffer loads used as they arrive (the scale sits in K), the key
// mask is the initial value of the last block's accumulator, the softmax is one pass (all keys present: true row max), P is normalised before it
// is packed (each lane owns one query's scores: no cross-lane fetch of 1 / sum), and the output goes out through buffer stores whose address is a
// loop-invariant per-lane offset (out of range for padded columns: dropped by the hardware) plus a wave-uniform row offset in an SGPR.
// PF (option flash_short = 2): the Q fragments of block it + 1 are requested before block it's MFMAs — a wave runs its blocks one after the other with
// one other wave on its SIMD, so without it every block starts with an exposed global-load latency.
typedef unsigned int fa_u32x4_t __attribute__((__vector_size__(4 * sizeof(unsigned int))));
template <int DKP, bool QF16, bool OUT16, bool PF = false>
__global__ __launch_bounds__(256, 2) void k_flash_short(FAArgs g) {
    constexpr int NDV  = 2;
    constexpr int KS   = DKP / 16;
    constexpr int KROW = DKP + 8;
    constexpr int DCH  = DKP / 8;
    constexpr int NKB  = 3;             // key blocks of 32
    constexpr int NKT  = 2 * NKB;       // P V k-steps of 16 keys
    constexpr int NK   = 32 * NKB;      // keys held
    constexpr int VRS  = fa_vtr_stride(NDV);
    constexpr int VCH  = NDV * 4;       // 8-wide chunks per V row (all columns of the accumulator, zero beyond D)
    constexpr uint32_t OOR = 0x7fffff00u;  // a per-lane buffer offset beyond every num_records: loads return 0, stores are dropped
    __shared__ __attribute__((aligned(16))) _Float16 Ks[NK * KROW];
    __shared__ __attribute__((aligned(16))) _Float16 Vs[NK * VRS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi   = lane >> 5;
    const int hn   = blockIdx.y;
    const char* kbase = g.k + (int64_t)hn * g.k_nb2;
    const char* vbase = g.v + (int64_t)hn * g.v_nb2;
    const half8_t z8  = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = threadIdx.x; e < NK * DCH; e += 256) {
        const int key = e / DCH, ch = e - key * DCH;
        half8_t kv    = z8;
        if (key < g.Lk && ch * 8 < g.D) {
            kv = *(const half8_t*)(kbase + (int64_t)key * g.k_nb1 + ch * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) kv[j] = (_Float16)((float)kv[j] * g.scale_log2e);  // scores come out of the MFMA in log2 units
        }
        *(half8_t*)&Ks[key * KROW + ch * 8] = kv;
    }
    for (int e = threadIdx.x; e < NK * VCH; e += 256) {
        const int key = e / VCH, ch = e - key * VCH;
        half8_t vv    = z8;  // rows beyond Lk and columns beyond D are zeros: their P is 0, their products must stay finite
        if (key < g.Lk && ch * 8 < g.DV) vv = *(const half8_t*)(vbase + (int64_t)key * g.v_nb1 + ch * 16);
        *(half8_t*)&Vs[key * VRS + ch * 8] = vv;
    }
    __syncthreads();
    half8_t kf[NKB][KS], vf[NKT][NDV];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const half8_t*)&Ks[(kb * 32 + (lane & 31)) * KROW + ks * 16 + hi * 8];
    const int vtr_lane = (4 * hi + ((lane & 15) >> 2)) * VRS + ((lane >> 4) & 1) * 16 + 4 * (lane & 3);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int nb = 0; nb < NDV; ++nb) {
            const _Float16* p = Vs + vtr_lane + t * 16 * VRS + nb * 32;
            const half4_t a = lds_read_tr16(p), c = lds_read_tr16(p + 8 * VRS);
            vf[t][nb] = (half8_t){a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        }
    // this lane holds query (lane & 31) and, of key block kb, the keys kb*32 + (r&3) + 8*(r>>2) + 4*hi.  Keys >= Lk exist in the last block only
    // (64 < Lk <= 96): -inf as the accumulator's initial value masks them without an instruction in the loop
    float16_t negc;
#pragma unroll
    for (int r = 0; r < 16; ++r) negc[r] = (64 + (r & 3) + 8 * (r >> 2) + 4 * hi >= g.Lk) ? -INFINITY : 0.f;

    // ---- loop-invariant addressing
    constexpr int QES = QF16 ? 2 : 4;
    const char* qhead = g.q + (int64_t)hn * g.q_nb2;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)qhead, 0, (int)min((int64_t)(g.Lq - 1) * g.q_nb1 + (int64_t)g.D * QES, (int64_t)0x7ffffff0), 0x00020000);
    const int hh = g.H > 0 ? hn % g.H : hn, nn = g.H > 0 ? hn / g.H : 0;
    char* obytes;
    int64_t ostride;
    if (OUT16) {
        obytes  = (char*)(g.dst16 + (int64_t)nn * g.Lq * g.ld16 + (int64_t)hh * g.DV);
        ostride = g.ld16 * 2;
    } else {
        obytes  = (char*)g.dst + (int64_t)hh * g.dst_nb_h + (int64_t)nn * g.dst_nb_n;
        ostride = g.dst_nb_q;
    }
    constexpr int OES = OUT16 ? 2 : 4;
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)obytes, 0, (int)min((int64_t)(g.Lq - 1) * ostride + (int64_t)g.DV * OES, (int64_t)0x7ffffff0), 0x00020000);
    uint32_t ovoff[NDV];  // row 4*hi of a block, column nb*32 + (lane & 31); padded columns point out of range
#pragma unroll
    for (int nb = 0; nb < NDV; ++nb) {
        const int d = nb * 32 + (lane & 31);
        ovoff[nb]   = d < g.DV ? (uint32_t)(4 * hi) * (uint32_t)ostride + (uint32_t)d * OES : OOR;
    }
    const int qwg = 128 * g.qi;  // queries per workgroup
    // raw Q registers of one block: Q[q][ks*16 + hi*8 .. +8], chunks beyond D read as zeros (offset out of range); a ragged last block reads its
    // last valid row again (those results are not stored)
    constexpr int NQR = QF16 ? KS : 2 * KS;
    fa_u32x4_t qraw[NQR];
#define FS_LOADQ(Q0_)                                                                                                                   \
    {                                                                                                                                   \
        const int rowl_  = min(lane & 31, g.Lq - 1 - (Q0_));                                                                            \
        const int soffq_ = (Q0_) * (int)g.q_nb1;                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                                             \
            const int d0        = ks * 16 + hi * 8;                                                                                     \
            const uint32_t voff = d0 < g.D ? (uint32_t)rowl_ * (uint32_t)g.q_nb1 + (uint32_t)d0 * QES : OOR;                            \
            if constexpr (QF16) {                                                                                                       \
                qraw[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (int)voff, soffq_, 0);                                            \
            } else {                                                                                                                    \
                qraw[2 * ks]     = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (int)voff, soffq_, 0);                                    \
                qraw[2 * ks + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (int)(voff == OOR ? OOR : voff + 16u), soffq_, 0);        \
            }                                                                                                                           \
        }                                                                                                                               \
    }
    if constexpr (PF) {
        const int q00 = blockIdx.x * qwg + wave * 32;
        if (q00 < g.Lq) FS_LOADQ(q00)
    }
    for (int it = 0; it < g.qi; ++it) {
        const int q0 = blockIdx.x * qwg + (it * 4 + wave) * 32;  // wave-uniform
        if (q0 >= g.Lq) break;
        const bool full  = q0 + 32 <= g.Lq;
        if constexpr (!PF) FS_LOADQ(q0)
        // ---- Q fragments (B operand of S^T = K Q^T) as f16
        half8_t qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (QF16) {
                qf[ks] = __builtin_bit_cast(half8_t, qraw[ks]);
            } else {
                const float4 a = __builtin_bit_cast(float4, qraw[2 * ks]), c = __builtin_bit_cast(float4, qraw[2 * ks + 1]);
                const half2_t h0 = __builtin_convertvector((float2_t){a.x, a.y}, half2_t), h1 = __builtin_convertvector((float2_t){a.z, a.w}, half2_t);
                const half2_t h2 = __builtin_convertvector((float2_t){c.x, c.y}, half2_t), h3 = __builtin_convertvector((float2_t){c.z, c.w}, half2_t);
                qf[ks] = (half8_t){h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
            }
        }
        if constexpr (PF) {  // the next block's rows: in flight during this block's MFMAs, softmax and stores
            const int q1 = q0 + 128;
            if (it + 1 < g.qi && q1 < g.Lq) FS_LOADQ(q1)
        }
        float16_t sc[NKB];
        sc[0] = (float16_t){0};
        sc[1] = (float16_t){0};
        sc[2] = negc;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[ks], sc[kb], 0, 0, 0);
        float m = sc[0][0];  // key 0 (block 0, r = 0) is never masked for hi = 0; the other half's partial max is merged below
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            m = __builtin_fmaxf(__builtin_fmaxf(m, sc[0][r]), sc[1][r]);  // v_max3_f32
            m = __builtin_fmaxf(m, sc[2][r]);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[kb][r] = __builtin_amdgcn_exp2f(sc[kb][r] - m);
                psum += sc[kb][r];
            }
        const float inv = 1.0f / (psum + __shfl_xor(psum, 32, 64));  // >= one term equal to 1
        half8_t pa[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            const int kb = t >> 1, rb = (t & 1) * 8;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const half2_t h2 = __builtin_convertvector((float2_t){sc[kb][rb + j] * inv, sc[kb][rb + j + 1] * inv}, half2_t);
                pa[t][j]     = h2[0];
                pa[t][j + 1] = h2[1];
            }
        }
        float16_t o[NDV];
#pragma unroll
        for (int nb = 0; nb < NDV; ++nb) o[nb] = (float16_t){0};
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int nb = 0; nb < NDV; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[t], vf[t][nb], o[nb], 0, 0, 0);
        // ---- rows (r&3) + 8*(r>>2) + 4*hi of the block, column nb*32 + (lane & 31).  Written as plain unrolled loops: behind a lambda taking the
        // register index as a run-time argument hipcc (ROCm 7.2) stored element 0 of the accumulator sixteen times (seen in the ISA, found on the GPU)
#define FS_PUT(VO_)                                                                                                             \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                                \
        const int ro   = (r & 3) + 8 * (r >> 2);                                                                                    \
        const int soff = (q0 + ro) * (int)ostride; /* wave-uniform */                                                               \
        _Pragma("unroll") for (int nb = 0; nb < NDV; ++nb) {                                                                        \
            const uint32_t vo = (VO_);                                                                                              \
            const float val   = o[nb][r];                                                                                           \
            if constexpr (OUT16)                                                                                                    \
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)val), rsO, (int)vo, soff, 0);             \
            else                                                                                                                    \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rsO, (int)vo, soff, 0);                         \
        }                                                                                                                           \
    }
        if (full) {  // wave-uniform: every row of the block exists
            FS_PUT(ovoff[nb])
        } else {
            FS_PUT(q0 + ro + 4 * hi < g.Lq ? ovoff[nb] : OOR)
        }
#undef FS_PUT
    }
#undef FS_LOADQ
}

// =====================================================================================================================================
// k_flash_pp — the PING-PONG kernel (round 3).  Why: in k_flash_attn every wave runs QK^T (MFMA) -> softmax (VALU) -> PV (MFMA) as one dependent
// chain and all four waves of a workgroup move through those phases together; the counters (profiles/r03c_pmc_flash.txt) show the consequence:
// matrix-pipe time + VALU time + LDS time add up to the whole kernel — the two pipes hardly ever work at the same moment.  Two query blocks per
// wave (QB = 2 above) gave 6 %.  Here a workgroup has EIGHT waves = two groups of four (waves w and w + 4 share a SIMD) whose tile loops are
// offset by half an iteration and kept there by the workgroup barrier: while group X issues the MFMAs of its tile (S(t) = K(t) Q^T and
// O += P(t-1) V(t-1), back to back, operands in LDS / registers), group Y does its VALU half (softmax of S -> P, the staging of the next K / V
// half-tiles, the next global loads), then they swap.  On every SIMD one wave feeds the matrix pipe while its partner feeds the VALU — by
// construction, not by luck.  The PV product is software-pipelined one tile behind QK^T, so a wave's MFMA phase never waits for its own softmax.
//
//   phase p:   even p: X = MFMA(t = p/2),     Y = VALU(t = p/2 - 1)  (+ staging)        one s_barrier at the end of every phase
//              odd  p: X = VALU(t = (p-1)/2), Y = MFMA(t = (p-1)/2)
// Staging: the group in its VALU phase stages one HALF (X: keys 32..63, Y: keys 0..31) of K(p/2 + 1) and of V(p/2) from registers loaded FOUR
// phases earlier (T14 split, two register sets), K and V double-buffered in LDS.  Life times (T = tile): K(T) is written in phases 2T-2, 2T-1 and read in 2T, 2T+1;
// V(T) is written in 2T, 2T+1 and read in 2T+2, 2T+3; the buffer it replaces (T-2) was last read in phase 2T-1 resp. 2T-3.
// Per query the arithmetic and its order are those of k_flash_attn (same MFMA shapes, deferred max, max slot, ones column): results are bit-identical.
template <int DKP, int NDV, bool MSLOT>
__global__ __launch_bounds__(512, 2) void k_flash_pp(FAArgs g) {
    constexpr int KS     = DKP / 16;
    constexpr int KROW   = DKP + 8;
    constexpr int DCH    = DKP / 8;
    constexpr int TILE_H = FA_KT * KROW + NDV * 32 * FA_VTS;
    constexpr int NCHP   = (32 * DCH + 255) / 256;  // chunks per thread per staging phase (half a tile by 256 threads; K and V each)
    constexpr int QWG    = 256;
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * TILE_H];
    _Float16* Ks = smem;
    _Float16* Vt = smem + FA_KT * KROW;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp  = wave >> 2;  // 0: group X (MFMA in even phases), 1: group Y
    const int hi   = lane >> 5;
    const int tg   = threadIdx.x & 255;  // thread index inside its group (staging)
    int hn, qb;
    if (g.grp > 0) {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int h = j % g.grp, unit = (j / g.grp) * 8 + xcd;
        if (unit >= g.units) return;
        const int nrb = (g.Lq + QWG - 1) / QWG;
        hn = (unit / nrb) * g.grp + h;
        qb = unit % nrb;
    } else {
        hn = blockIdx.y;
        qb = blockIdx.x;
    }
    const int q0 = qb * QWG + wave * 32;
    const int qi = q0 + (lane & 31);

    // ---- Q fragments (as k_flash_attn): f16 image or f32 rows through LDS, or per-lane reads
    half8_t qf[KS];
    if (g.q_f16 || g.q_vec) {
        constexpr int QROW = DKP + 4;
        static_assert(QWG * QROW <= 2 * TILE_H, "Q staging must fit the K/V tile buffers");
        const char* qblk = g.q + (int64_t)hn * g.q_nb2 + (int64_t)(qb * QWG) * g.q_nb1;
        if (g.q_f16) {
            constexpr int C8 = DKP / 8;
            for (int e = threadIdx.x; e < QWG * C8; e += 512) {
                const int row = e / C8, c8 = e - row * C8;
                half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (qb * QWG + row < g.Lq && c8 * 8 < g.D) v = *(const half8_t*)(qblk + (int64_t)row * g.q_nb1 + c8 * 16);
                half4_t a, b;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = (_Float16)((float)v[j] * g.scale_log2e);
                    b[j] = (_Float16)((float)v[4 + j] * g.scale_log2e);
                }
                *(half4_t*)&smem[row * QROW + c8 * 8]     = a;
                *(half4_t*)&smem[row * QROW + c8 * 8 + 4] = b;
            }
        } else {
            constexpr int C4 = DKP / 4;
            for (int e = threadIdx.x; e < QWG * C4; e += 512) {
                const int row = e / C4, c4 = e - row * C4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (qb * QWG + row < g.Lq && c4 * 4 < g.D) v = *(const float4*)(qblk + (int64_t)row * g.q_nb1 + c4 * 16);
                half4_t h;
                h[0] = (_Float16)(v.x * g.scale_log2e);
                h[1] = (_Float16)(v.y * g.scale_log2e);
                h[2] = (_Float16)(v.z * g.scale_log2e);
                h[3] = (_Float16)(v.w * g.scale_log2e);
                *(half4_t*)&smem[row * QROW + c4 * 4] = h;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const _Float16* p = &smem[(wave * 32 + (lane & 31)) * QROW + ks * 16 + hi * 8];
            const half4_t a = *(const half4_t*)p, c = *(const half4_t*)(p + 4);
            qf[ks] = (half8_t){a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        }
        __syncthreads();
    } else {
        const float* qrow = (const float*)(g.q + (int64_t)min(qi, g.Lq - 1) * g.q_nb1 + (int64_t)hn * g.q_nb2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = ks * 16 + hi * 8 + j;
                qf[ks][j]   = (_Float16)((d < g.D && qi < g.Lq) ? qrow[d] * g.scale_log2e : 0.f);
            }
    }

    float16_t o[NDV];
#pragma unroll
    for (int nb = 0; nb < NDV; ++nb) o[nb] = (float16_t){0};
    float m_run = MSLOT ? 0.f : -INFINITY, l_run = 0.f;

    const char* kbase = g.k + (int64_t)hn * g.k_nb2;
    const char* vbase = g.v + (int64_t)hn * g.v_nb2;
    const int nd8           = g.D / 8;
    const bool has_ones     = g.DV < NDV * 32 && g.D == g.DV && g.D % 8 == 0;
    const bool ones_in_tile = has_ones && g.DV < DKP;
    const int NT            = (g.Lk + FA_KT - 1) / FA_KT;

    // ---- staging of this group's half tiles: thread tg handles chunks e = tg + 256 c of the 32 keys [32 * half, 32 * half + 32)
    const int half = grp == 0 ? 1 : 0;
    // two register sets: the loads of staging step n are consumed at step n + 2 = FOUR phases later (with one set / two phases the kernel waited for
    // its global loads in every VALU phase: 68 % of the wave cycles parked, profiles/r04c_pmc_flash_pp.txt)
    // (d >= 80: one set, two phases ahead — a second set does not fit the 256 registers next to the accumulators)
    constexpr int NSET = DKP <= 64 ? 2 : 1;
    half8_t kregA[NCHP], vregA[NCHP], kregB[NSET == 2 ? NCHP : 1], vregB[NSET == 2 ? NCHP : 1];
    uint32_t koff[NCHP], voff[NCHP];  // offsets are always readable addresses (dummies: the tile's first bytes, zeroed at the store)
    int kkey[NCHP], vkey_[NCHP];  // key inside the TILE, or FA_KT (never valid)
    bool kone[NCHP], vone[NCHP], live[NCHP];
#pragma unroll
    for (int c = 0; c < NCHP; ++c) {
        const int e   = tg + c * 256;
        const int key = 32 * half + e / DCH, ch = e % DCH;
        const int vkey = 32 * half + (e & 31), vch = e >> 5;
        live[c]  = e < 32 * DCH;
        kkey[c]  = (live[c] && ch < nd8) ? key : FA_KT;
        kone[c]  = live[c] && ch == nd8;
        vkey_[c] = (live[c] && vch < nd8) ? vkey : FA_KT;
        vone[c]  = live[c] && ones_in_tile && vch == nd8;
        koff[c]  = kkey[c] < FA_KT ? (uint32_t)key * (uint32_t)g.k_nb1 + (uint32_t)ch * 16u : 0u;
        voff[c]  = vkey_[c] < FA_KT ? (uint32_t)vkey * (uint32_t)g.v_nb1 + (uint32_t)vch * 16u : 0u;
    }
    // (the descriptors' words are wave-uniform by construction; said explicitly, because under SGPR pressure — the two-block d = 64 instantiation — the compiler kept them in
    // VGPRs and wrapped every tile's loads in waterfall loops)
    auto uni_ptr = [](const char* p) {
        const uint64_t u = (uint64_t)(uintptr_t)p;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi_ << 32) | lo);
    };
    kbase = uni_ptr(kbase);
    vbase = uni_ptr(vbase);
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)min((int64_t)g.Lk * g.k_nb1, (int64_t)0x7fffffff), 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)min((int64_t)g.Lk * g.v_nb1, (int64_t)0x7fffffff), 0x00020000);
    // registers <- global: K half of tile tk, V half of tile tv (either may lie beyond the last tile: skipped); LDS <- registers (which also inserts the
    // constant 1s of the max slot / ones row: touching a loaded register earlier would put a vmcnt(0) wait right behind its load).  Macros over the
    // NAMED register sets: passing the arrays to a lambda by reference put them in scratch memory.
#define FPP_GLOAD(KR_, VR_, TK_, TV_)                                                                \
    do { /* buffer loads: uniform descriptor, constant per-lane offsets, tile offset in an SGPR (see gload in k_flash_attn) */ \
        const int tk_ = (TK_), tv_ = (TV_);                                                          \
        if (tk_ < NT) {                                                                              \
            const int so_ = tk_ * FA_KT * (int)g.k_nb1;                                              \
            _Pragma("unroll") for (int c = 0; c < NCHP; ++c)                                         \
                KR_[c] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rsK, (int)koff[c], so_, 0)); \
        }                                                                                            \
        if (tv_ < NT) {                                                                              \
            const int so_ = tv_ * FA_KT * (int)g.v_nb1;                                              \
            _Pragma("unroll") for (int c = 0; c < NCHP; ++c)                                         \
                VR_[c] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rsV, (int)voff[c], so_, 0)); \
        }                                                                                            \
    } while (0)
#define FPP_LSTORE(KR_, VR_, TK_, TV_)                                                               \
    do {                                                                                             \
        const int tk_ = (TK_), tv_ = (TV_);                                                          \
        const half8_t z_ = {0, 0, 0, 0, 0, 0, 0, 0};                                                 \
        if (tk_ < NT) {                                                                              \
            _Float16* ks_ = Ks + (tk_ & 1) * TILE_H;                                                 \
            const int left_ = g.Lk - tk_ * FA_KT;                                                    \
            _Pragma("unroll") for (int c = 0; c < NCHP; ++c) if (live[c]) {                          \
                const int e_ = tg + c * 256;                                                         \
                half8_t kv_  = kkey[c] < left_ ? KR_[c] : z_;                                        \
                if (MSLOT && kone[c]) kv_[0] = (_Float16)1.0f; /* the max slot: K[key][D] = 1 */     \
                *(half8_t*)&ks_[(32 * half + e_ / DCH) * KROW + (e_ % DCH) * 8] = kv_;               \
            }                                                                                        \
        }                                                                                            \
        if (tv_ < NT) {                                                                              \
            _Float16* vt_ = Vt + (tv_ & 1) * TILE_H;                                                 \
            const int left_ = g.Lk - tv_ * FA_KT;                                                    \
            _Pragma("unroll") for (int c = 0; c < NCHP; ++c) if (live[c]) {                          \
                const int e_ = tg + c * 256;                                                         \
                const int vk_ = 32 * half + (e_ & 31), vc_ = e_ >> 5;                                \
                half8_t vv_   = vkey_[c] < left_ ? VR_[c] : z_;                                      \
                if (vone[c]) vv_[0] = (_Float16)1.0f; /* the ones row: PV accumulates the row sums */ \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) vt_[(vc_ * 8 + j) * FA_VTS + vk_] = vv_[j]; \
            }                                                                                        \
        }                                                                                            \
    } while (0)
    // staging step n of this group (n = 0, 1, ...): the halves of K(n + 1) and V(n) go to LDS from the set loaded two steps ago, which is then
    // re-loaded for step n + 2
    // The set is chosen at COMPILE time (the tile loops below are unrolled by two): a run-time choice made the compiler merge the two sets through
    // register copies, i.e. wait for the fresh loads on the spot.  With one set (NSET == 1) both macros use set A, two phases ahead.
#define FPP_STAGE_A(N_)                                                    \
    do {                                                                   \
        const int n_ = (N_);                                               \
        FPP_LSTORE(kregA, vregA, n_ + 1, n_);                              \
        FPP_GLOAD(kregA, vregA, n_ + 1 + NSET, n_ + NSET);                 \
    } while (0)
#define FPP_STAGE_B(N_)                                                    \
    do {                                                                   \
        const int n_ = (N_);                                               \
        if constexpr (NSET == 1) {                                         \
            FPP_LSTORE(kregA, vregA, n_ + 1, n_);                          \
            FPP_GLOAD(kregA, vregA, n_ + 2, n_ + 1);                       \
        } else {                                                           \
            FPP_LSTORE(kregB, vregB, n_ + 1, n_);                          \
            FPP_GLOAD(kregB, vregB, n_ + 3, n_ + 2);                       \
        }                                                                  \
    } while (0)

    // ---- prologue (all 512 threads): clear the never-staged V^T rows, the ones row where it sits beyond the staged rows, K(0) in full
    for (int b = 0; b < 2; ++b)
        for (int e = threadIdx.x; e < NDV * 32 * FA_VTS / 2; e += 512) ((uint32_t*)(Vt + b * TILE_H))[e] = 0u;
    __syncthreads();
    if (has_ones && !ones_in_tile && threadIdx.x < FA_KT * 2) Vt[(threadIdx.x >> 6) * TILE_H + g.DV * FA_VTS + (threadIdx.x & 63)] = (_Float16)1.0f;
    for (int e = threadIdx.x; e < FA_KT * DCH; e += 512) {
        const int key = e / DCH, ch = e - key * DCH;
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < g.Lk && ch < nd8) v = *(const half8_t*)(kbase + (int64_t)key * g.k_nb1 + ch * 16);
        if (MSLOT && ch == nd8) v[0] = (_Float16)1.0f;
        *(half8_t*)&Ks[key * KROW + ch * 8] = v;
    }
    FPP_GLOAD(kregA, vregA, 1, 0);  // what this group stages in its first (two) VALU(-like) phase(s)
    if constexpr (NSET == 2) FPP_GLOAD(kregB, vregB, 2, 1);
    __syncthreads();

    // ---- the two phase bodies
    float16_t s[2];
    half8_t pa[4];
    // MFMA phase of tile t: S = K(t) Q^T (do_qk) and O += P(t-1) V(t-1) (do_pv)
    auto mphase = [&](int t, bool do_qk, bool do_pv) {
        if constexpr (DKP <= 64) {
            // ALL LDS fragment reads of the phase (K for QK^T, V^T for PV) are requested before the first MFMA: the wave has nothing else to do in this
            // phase, so a read issued between MFMAs is a read whose latency the matrix pipe waits for (phases of ~1400 cycles against 450 cycles
            // of MFMA work, profiles/r04e_pmc_flash_pp.txt)
            half8_t kf[2][KS], vf[4][NDV];
            if (do_qk) {
                const _Float16* Kc = Ks + (t & 1) * TILE_H;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const half8_t*)&Kc[(kb * 32 + (lane & 31)) * KROW + ks * 16 + hi * 8];
            }
            if (do_pv) {
                const _Float16* Vc = Vt + ((t - 1) & 1) * TILE_H;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int nb = 0; nb < NDV; ++nb) {
                        const _Float16* vrow = &Vc[(nb * 32 + (lane & 31)) * FA_VTS + tt * 16 + 4 * hi];
                        const half4_t v0 = *(const half4_t*)vrow, v1 = *(const half4_t*)(vrow + 8);
                        vf[tt][nb]       = (half8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_pv) {  // P(t-1) has been in registers since the last VALU phase: these MFMAs wait for V fragments only
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int nb = 0; nb < NDV; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[tt], vf[tt][nb], o[nb], 0, 0, 0);
            }
            if (do_qk) {
                s[0] = (float16_t){0};
                s[1] = (float16_t){0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
            }
            return;
        }
        if (do_qk) {
            const _Float16* Kc = Ks + (t & 1) * TILE_H;
            if constexpr (KS <= 6) {
                half8_t kf[2][KS];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const half8_t*)&Kc[(kb * 32 + (lane & 31)) * KROW + ks * 16 + hi * 8];
                s[0] = (float16_t){0};
                s[1] = (float16_t){0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    s[kb] = (float16_t){0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const half8_t kf = *(const half8_t*)&Kc[(kb * 32 + (lane & 31)) * KROW + ks * 16 + hi * 8];
                        s[kb]            = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
                    }
                }
            }
        }
        if (do_pv) {
            const _Float16* Vc = Vt + ((t - 1) & 1) * TILE_H;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int nb = 0; nb < NDV; ++nb) {
                    const _Float16* vrow = &Vc[(nb * 32 + (lane & 31)) * FA_VTS + tt * 16 + 4 * hi];
                    const half4_t v0 = *(const half4_t*)vrow, v1 = *(const half4_t*)(vrow + 8);
                    const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[nb]            = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[tt], vf, o[nb], 0, 0, 0);
                }
        }
    };
    // VALU phase of tile t: online softmax of S(t) -> P(t) in f16 (the A operand of the next PV), running max / sum, rare rescale of O
    auto vphase = [&](int t) {
        const int kt = t * FA_KT;
        if (kt + FA_KT > g.Lk) {  // ragged last tile: mask keys >= Lk
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= g.Lk) s[kb][r] = -INFINITY;
        }
        float tmax = s[0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, s[0][r]), s[1][r]);
        if (MSLOT ? (t == 0 || __any(tmax > FA_THR)) : __any(tmax > m_run + FA_THR)) {
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            float alpha;
            if constexpr (MSLOT) {
                const float m_new = fminf((float)(_Float16)fminf(m_run + (t == 0 ? tmax : fmaxf(tmax, 0.f)), 65504.f), 65504.f);
                const float delta = m_new - m_run;
                alpha             = __builtin_amdgcn_exp2f(-delta);
                m_run             = m_new;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
                if (hi) qf[KS - 1][0] = (_Float16)(-m_new);
            } else {
                const float m_new = fmaxf(m_run, tmax);
                alpha             = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run             = m_new;
            }
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row  = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float ar = __shfl(alpha, row, 64);
#pragma unroll
                for (int nb = 0; nb < NDV; ++nb) o[nb][r] *= ar;
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(MSLOT ? s[kb][r] : s[kb][r] - m_run);
        if (!has_ones) {
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) psum += s[kb][r];
            l_run += psum;
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int kb = tt >> 1, rb = (tt & 1) * 8;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const half2_t h2 = __builtin_convertvector((float2_t){s[kb][rb + j], s[kb][rb + j + 1]}, half2_t);
                pa[tt][j]     = h2[0];
                pa[tt][j + 1] = h2[1];
            }
        }
    };

    // ---- the two group programs (2 NT barriers each; see the phase table above).  The phase barrier waits for this wave's LDS traffic only:
    // __syncthreads() also drains vmcnt, i.e. it waited for the global loads issued a moment earlier for a LATER phase — every VALU phase then
    // lasted one L2 round trip (phases of ~2300 cycles instead of ~700, 69 % of the wave cycles parked: profiles/r04d_pmc_flash_pp.txt)
    // (sched_barrier on both sides: the asm only orders MEMORY operations — without them the compiler moved most of a phase's MFMAs behind the barrier)
#define FPP_BARRIER()                                                         \
    do {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                                    \
    } while (0)
    if (grp == 0) {
        int t = 0;
        for (; t + 1 < NT; t += 2) {
            mphase(t, true, t > 0);   // phase 2t
            FPP_BARRIER();
            FPP_STAGE_A(t);           // phase 2t + 1: K(t+1), V(t) halves (staging step n = t: even -> set A)
            vphase(t);
            FPP_BARRIER();
            mphase(t + 1, true, true);
            FPP_BARRIER();
            FPP_STAGE_B(t + 1);
            vphase(t + 1);
            FPP_BARRIER();
        }
        if (t < NT) {
            mphase(t, true, t > 0);
            FPP_BARRIER();
            FPP_STAGE_A(t);
            vphase(t);
            FPP_BARRIER();
        }
        mphase(NT, false, true);      // phase 2 NT: the last PV
    } else {
        FPP_STAGE_A(0);               // phase 0: staging only (step n = 0)
        FPP_BARRIER();
        int t = 0;
        for (; t + 1 < NT; t += 2) {
            mphase(t, true, t > 0);   // phase 2t + 1
            FPP_BARRIER();
            FPP_STAGE_B(t + 1);       // phase 2t + 2: K(t+2), V(t+1) halves (step n = t + 1: odd -> set B)
            vphase(t);
            FPP_BARRIER();
            mphase(t + 1, true, true);
            FPP_BARRIER();
            FPP_STAGE_A(t + 2);
            vphase(t + 1);
            if (t + 2 < NT) FPP_BARRIER();
        }
        if (t < NT) {
            mphase(t, true, t > 0);
            FPP_BARRIER();
            FPP_STAGE_B(t + 1);
            vphase(t);
        }
        mphase(NT, false, true);      // phase 2 NT + 1
    }
#undef FPP_BARRIER

    // ---- finalise (as k_flash_attn)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv   = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int nb_l = g.DV >> 5, lane_l = (g.DV & 31) + 32 * hi;
    const int hh = g.H > 0 ? hn % g.H : hn, nn = g.H > 0 ? hn / g.H : 0;
    char* obase       = g.dst ? (char*)g.dst + (int64_t)hh * g.dst_nb_h + (int64_t)nn * g.dst_nb_n : nullptr;
    _Float16* obase16 = g.dst16 ? g.dst16 + (int64_t)nn * g.Lq * g.ld16 + (int64_t)hh * g.DV : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float ir      = __shfl(inv, row, 64);
        if (has_ones) {
            float lsum = 0.f;
#pragma unroll
            for (int nb = 0; nb < NDV; ++nb)
                if (nb == nb_l) lsum = o[nb][r];
            lsum = __shfl(lsum, lane_l, 64);
            ir   = lsum > 0.f ? 1.0f / lsum : 0.f;
        }
        const int q = q0 + row;
        if (q >= g.Lq) continue;
#pragma unroll
        for (int nb = 0; nb < NDV; ++nb) {
            const int d = nb * 32 + (lane & 31);
            if (d >= g.DV) continue;
            const float val = o[nb][r] * ir;
            if (obase) *(float*)(obase + (int64_t)q * g.dst_nb_q + d * 4) = val;
            if (obase16) obase16[(int64_t)q * g.ld16 + d] = (_Float16)val;
        }
    }
}

bool flash_attn_supported(int64_t D, int64_t DV) { return D == DV && D >= 8 && D <= 160; }

#ifdef MI355X_EXPERIMENTS  // wrong-result timing ablations: never part of the shipped library (build with -DMI355X_EXPERIMENTS)
static int g_flash_ablate = 0;
void flash_attn_set_ablate(int v) { g_flash_ablate = v; }
#endif

static int g_flash_mslot64 = 0;  // option "flash_mslot64" (launch_flash_attn)
void flash_attn_set_mslot64(int v) { g_flash_mslot64 = v; }
static int g_flash_mslot = 1;  // option "flash_mslot": 0 = subtract the running max on the VALU (A/B measurements)
void flash_attn_set_mslot(int v) { g_flash_mslot = v; }
static int g_flash_grid = 1;  // option "flash_grid": 0 = plain (query block, head) grid (A/B measurements)
void flash_attn_set_grid(int v) { g_flash_grid = v; }
static int g_flash_qb2 = 1;  // option "flash_qb2": 0 = one query block per wave everywhere (the round-2 kernel; A/B measurements)
void flash_attn_set_qb2(int v) { g_flash_qb2 = v; }
static int g_flash_pp = 0;  // option "flash_pp": 0 (default) = never the ping-pong kernel, 1 = for 64 < d <= 96, 2 = wherever it is legal (A/B measurements)
void flash_attn_set_pp(int v) { g_flash_pp = v; }
static int g_flash_vpf = 31;  // option "flash_vpf": head-dim classes (1: d <= 48, 2: <= 64, 4: <= 96, 8: <= 128, 16: above) whose kernel prefetches its fragments
void flash_attn_set_vpf(int v) { g_flash_vpf = v; }
static int g_flash_vtr = 31;  // option "flash_vtr": head-dim classes (bits as flash_vpf) whose prefetching kernel keeps V row-major in LDS and reads it with ds_read_b64_tr_b16 (0 = the transposed tile of rounds 1-3)
void flash_attn_set_vtr(int v) { g_flash_vtr = v; }
static int g_flash_ovl = 1;  // option "flash_ovl": 1 = the two-block d = 40 kernel with one block's softmax issued inside the other block's MFMAs; 2 = also the other d <= 48 launches (shapes no supported model has; checked by scripts/flash_check.py ovl2 only); 0 = phase-by-phase order
void flash_attn_set_ovl(int v) { g_flash_ovl = v; }
static int g_flash_nsel = 1;  // option "flash_nsel": 1 = select-free staging in the d = 40 two-block, d = 64 and d = 128 kernels (round 4: bit-identical, d = 128 346 -> 326 us, SD1.5 step -0.6 %; profiles/r05a_*)
void flash_attn_set_nsel(int v) { g_flash_nsel = v; }
static int g_flash_short = 2;  // option "flash_short": k_flash_short for 64 < Lk <= 96, d <= 64 (the 77-token cross-attentions): 1 = on, 2 = with the next block's Q rows prefetched (default; round 4: 81 -> 40 us per SD1.5 64x64-level launch), 0 = the tile kernel
void flash_attn_set_short(int v) { g_flash_short = v; }
static int g_flash_pk = 0;  // option "flash_pk": 1 = max subtraction and row sums as packed f32 operations in the d = 64 / d = 128 kernels (k_flash_attn PK; measured SLOWER: profiles/r09a_flash_pk_rejected.txt), 0 = scalar
void flash_attn_set_pk(int v) { g_flash_pk = v; }
static int g_flash_sm = 0;  // option "flash_sm": softmax arithmetic variant of the d = 64 / d = 128 one-block kernels (k_flash_attn SM bits: 2 = MINIT, 4 = DOT2, 6 = both); flash_pk = 1 is bit 1
void flash_attn_set_sm(int v) { g_flash_sm = v; }
static int g_flash_qb64 = 0;  // option "flash_qb64": N > 0 = two query blocks per wave also at d = 64 when the launch has at least N workgroups of 256 queries (prefetching row-major-V kernel); 0 = off
void flash_attn_set_qb64(int v) { g_flash_qb64 = v; }
static int g_flash_pp_min_tiles = 4;  // option "flash_pp_min_tiles": key tiles (64 keys) from which the ping-pong pipeline has a steady state worth its prologue
void flash_attn_set_pp_min_tiles(int v) { g_flash_pp_min_tiles = v; }

void launch_flash_attn(hipStream_t s, const FlashOut& out, const View4& q, const View4& k, const View4& v, float scale) {
    KScope ks_(s, KF_FLASH, 4.0 * (double)q.ne[1] * (double)k.ne[1] * (double)q.ne[2] * (double)q.ne[0], 0.0);  // 4 * Lq * Lk * (H*N) * d
    FAArgs g;
    g.qi = 0;
    g.q = (const char*)q.data;
    g.k = (const char*)k.data;
    g.v = (const char*)v.data;
    g.dst      = out.dst;
    g.dst16    = (_Float16*)out.dst16;
    g.ld16     = out.ld16;
    g.H        = out.H;
    g.dst_nb_n = out.nb_n;
    const int64_t dst_nb_q = out.nb_q, dst_nb_h = out.nb_h;
    g.q_nb1 = q.nb[1];
    g.q_nb2 = q.nb[2];
    g.k_nb1 = k.nb[1];
    g.k_nb2 = k.nb[2];
    g.v_nb0 = v.nb[0];
    g.v_nb1 = v.nb[1];
    g.v_nb2 = v.nb[2];
    g.dst_nb_q = dst_nb_q;
    g.dst_nb_h = dst_nb_h;
    g.Lq = (int)q.ne[1];
    g.Lk = (int)k.ne[1];
    g.D  = (int)q.ne[0];
    g.DV = (int)v.ne[0];
    g.kv_f16      = k.type == 1;
    g.scale_log2e = scale * 1.44269504088896340736f;
    auto al16 = [](const void* p, int64_t a, int64_t b) { return ((((uintptr_t)p) | (uintptr_t)a | (uintptr_t)b) & 15) == 0; };
    g.vec_ok = (g.D % 8 == 0) && al16(k.data, k.nb[1], k.nb[2]) && al16(v.data, v.nb[1], v.nb[2]);
    g.q_vec = q.type == 0 && (g.D % 4 == 0) && q.nb[0] == 4 && al16(q.data, q.nb[1], q.nb[2]);
    g.q_f16 = q.type == 1;
    if (g.q_f16 && !(g.D % 8 == 0 && q.nb[0] == 2 && al16(q.data, q.nb[1], q.nb[2]))) {
        fprintf(stderr, "mi355x: flash attention: f16 Q must be d-contiguous, 16-byte aligned, d %% 8 == 0\n");
        abort();
    }
    const int D     = g.D;
    const bool fast = g.vec_ok && g.kv_f16 && v.nb[0] == 2 && k.nb[0] == 2 && g.D == g.DV;
    if (g_flash_short && fast && D <= 64 && g.Lk <= 96 && g.Lk > 64 && (g.q_f16 || g.q_vec) && g.D % 8 == 0 && (g.dst != nullptr) != (g.dst16 != nullptr) &&
        (int64_t)g.Lq * std::max<int64_t>(g.q_nb1, g.dst16 ? g.ld16 * 2 : dst_nb_q) < (int64_t)0x7ff00000) {
        // blocks of 32 queries per wave: enough to amortise the K / V fragment set-up, few enough to leave every CU >= 2 rounds of workgroups
        const int64_t nblk = (int64_t)((g.Lq + 31) / 32) * q.ne[2];
        g.qi  = (int)std::max<int64_t>(1, std::min<int64_t>(8, nblk / 4096));
        g.grp = g.units = 0;
        dim3 gs((unsigned)((g.Lq + 128 * g.qi - 1) / (128 * g.qi)), (unsigned)q.ne[2]);
#define FS_CASE(DKP_, PF_)                                                     \
    do {                                                                       \
        if (g.q_f16 && g.dst16)                                                \
            k_flash_short<DKP_, true, true, PF_><<<gs, 256, 0, s>>>(g);        \
        else if (g.q_f16)                                                      \
            k_flash_short<DKP_, true, false, PF_><<<gs, 256, 0, s>>>(g);       \
        else if (g.dst16)                                                      \
            k_flash_short<DKP_, false, true, PF_><<<gs, 256, 0, s>>>(g);       \
        else                                                                   \
            k_flash_short<DKP_, false, false, PF_><<<gs, 256, 0, s>>>(g);      \
    } while (0)
        if (g_flash_short >= 2) {
            if (D <= 48)
                FS_CASE(48, true);
            else
                FS_CASE(64, true);
        } else {
            if (D <= 48)
                FS_CASE(48, false);
            else
                FS_CASE(64, false);
        }
#undef FS_CASE
        return;
    }
    // two query blocks per wave (256 queries per workgroup) when the launch still gives every CU two workgroups' worth of work
    const int64_t wg256 = ((int64_t)(g.Lq + 255) / 256) * q.ne[2];
    const int NT        = (g.Lk + FA_KT - 1) / FA_KT;
    // the ping-pong kernel (8 waves = 256 queries per workgroup, one workgroup per CU): long key loops on grids that fill their rounds.  Measured
    // (profiles/r04f_flash_variants.txt, HIP events, one box): it wins at d = 80 (90.6 vs 97.6 us at L = 1024), ties at d = 128 and LOSES at
    // d <= 64 (d = 40, L = 4096: 814 vs 657 us for two query blocks per wave) — the counters show matrix-pipe time + VALU time + LDS time still
    // adding up to the kernel time although the two groups are in opposite phases by construction: on this part a SIMD does not overlap one
    // wave's MFMAs with its partner's VALU stream the way the phase picture assumes, so the extra barriers are pure cost.  Inside the SD1.5 forward
    // even the d = 80 win does not survive (profiles/r04g_ab_flash_pp.txt: step 25.70 ms without it, 25.87 with it for d = 80, 26.60 everywhere):
    // default OFF; flash_pp = 1 takes it for d in (64, 96], 2 wherever it is legal (A/B runs)
    const bool pp_ok    = fast && D <= 128 && NT >= g_flash_pp_min_tiles && g.Lq >= 192 && wg256 >= 256 && wg256 * 5 >= ((wg256 + 255) / 256) * 256 * 4;
    const bool pp       = pp_ok && (g_flash_pp == 2 || (g_flash_pp == 1 && D > 64 && D <= 96));
    // two query blocks per wave: d = 40 only (d = 64 measured slower, d >= 80 spills).  Other head dims <= 48 on grids this large occur in none of the
    // supported models and no test reaches them, so by default they stay on the one-block kernels every test runs; flash_qb2 = 2 sends them here too
    const bool qb2      = !pp && g_flash_qb2 && fast && D <= 48 && (g_flash_qb2 >= 2 || (D == 40 && g_flash_mslot)) && NT >= 4 && g.Lq >= 192 && wg256 >= 512;
    // d = 64 (round 6): the same two-block structure on the prefetching row-major-V kernel, where the launch leaves enough workgroups of 256 queries
    const bool qb64     = !pp && g_flash_qb64 > 0 && fast && D == 64 && g_flash_nsel && (g_flash_vpf & 2) && (g_flash_vtr & 2) && NT >= 4 && g.Lq >= 192 && wg256 >= g_flash_qb64;
    const int QWG       = (qb2 || pp || qb64) ? 256 : 128;
    dim3 grid((unsigned)((g.Lq + QWG - 1) / QWG), (unsigned)q.ne[2]);
    g.grp = g.units = 0;
    if (g_flash_grid && out.H > 0 && q.ne[2] % out.H == 0) {
        g.grp   = out.H;
        g.units = (int)(q.ne[2] / out.H) * (int)grid.x;
        grid    = dim3((unsigned)(((g.units + 7) / 8) * 8 * g.grp), 1u);
    }
#define FA_CASE(DKP_, NDV_)                                                                \
    do {                                                                                   \
        constexpr int cls_ = DKP_ <= 48 ? 1 : DKP_ <= 64 ? 2 : DKP_ <= 96 ? 4 : DKP_ <= 128 ? 8 : 16;  \
        if (fast && (g_flash_vpf & cls_) && (g_flash_vtr & cls_))                          \
            k_flash_attn<DKP_, NDV_, true, 0, false, 1, true, true><<<grid, 256, 0, s>>>(g);  \
        else if (fast && (g_flash_vpf & cls_))                                             \
            k_flash_attn<DKP_, NDV_, true, 0, false, 1, true><<<grid, 256, 0, s>>>(g);     \
        else if (fast)                                                                     \
            k_flash_attn<DKP_, NDV_, true><<<grid, 256, 0, s>>>(g);                        \
        else                                                                               \
            k_flash_attn<DKP_, NDV_, false><<<grid, 256, 0, s>>>(g);                       \
    } while (0)
#ifdef MI355X_EXPERIMENTS
    if (D <= 48 && fast && g_flash_ablate == 1) {
        k_flash_attn<48, 2, true, 1><<<grid, 256, 0, s>>>(g);
        return;
    }
    if (D <= 48 && fast && g_flash_ablate == 2) {
        k_flash_attn<48, 2, true, 2><<<grid, 256, 0, s>>>(g);
        return;
    }
#endif
    if (pp) {
        if (D == 40 && g_flash_mslot)
            k_flash_pp<48, 2, true><<<grid, 512, 0, s>>>(g);
        else if (D <= 48)
            k_flash_pp<48, 2, false><<<grid, 512, 0, s>>>(g);
        else if (D <= 64)
            k_flash_pp<64, 2, false><<<grid, 512, 0, s>>>(g);
        else if (D <= 80)
            k_flash_pp<80, 3, false><<<grid, 512, 0, s>>>(g);
        else if (D <= 96)
            k_flash_pp<96, 3, false><<<grid, 512, 0, s>>>(g);
        else
            k_flash_pp<128, 4, false><<<grid, 512, 0, s>>>(g);
        return;
    }
    if (qb2 && (g_flash_vpf & 1) && (g_flash_vtr & 1) && g_flash_ovl && D == 40 && g_flash_mslot) {
        if (g_flash_nsel)
            k_flash_attn<48, 2, true, 0, true, 2, true, true, true, true><<<grid, 256, 0, s>>>(g);
        else
            k_flash_attn<48, 2, true, 0, true, 2, true, true, true><<<grid, 256, 0, s>>>(g);
        return;
    }
    if (g_flash_nsel && !qb2 && !pp && fast && (D == 64 || D == 128) && (g_flash_vpf & (D == 64 ? 2 : 8)) && (g_flash_vtr & (D == 64 ? 2 : 8))) {
        // d = 64 is bound by instruction ISSUE (issue port 99.5 % busy, matrix pipe 40 %: profiles/r05g_pmc_sq_flash.txt).  Measured and REJECTED (round 4,
        // profiles/r05h_flash_mslot64_rejected.txt; -DMI355X_EXPERIMENTS builds keep it behind option "flash_mslot64"): trading matrix work for VALU work — the
        // running max in a fifth k-step (80-wide tile: +2 MFMAs per tile, -32 v_sub per lane) is 25-27 % SLOWER (L = 4096: 139 -> 175 us; L = 4250: 534 ->
        // 630-660 us), with the row sums in a ones column of a third V block on top (+4 MFMAs, -32 v_add) 30 % slower: once more, matrix-pipe time and VALU
        // time ADD on this part; the d = 40 max slot pays only because its k-step and its V column were padding anyway
#ifdef MI355X_EXPERIMENTS
        if (D == 64 && g_flash_mslot64 == 1)
            k_flash_attn<80, 2, true, 0, true, 1, true, true, false, true><<<grid, 256, 0, s>>>(g);
        else if (D == 64 && g_flash_mslot64 == 2)
            k_flash_attn<80, 3, true, 0, true, 1, true, true, false, true><<<grid, 256, 0, s>>>(g);
        else
#endif
        if (D == 64 && qb64 && g_flash_pk)
            k_flash_attn<64, 2, true, 0, false, 2, true, true, false, true, 1><<<grid, 256, 0, s>>>(g);
        else if (D == 64 && qb64)
            k_flash_attn<64, 2, true, 0, false, 2, true, true, false, true><<<grid, 256, 0, s>>>(g);
        else if (D == 64 && g_flash_pk)
            k_flash_attn<64, 2, true, 0, false, 1, true, true, false, true, 1><<<grid, 256, 0, s>>>(g);
        else if (D == 64 && g_flash_sm == 2)
            k_flash_attn<64, 2, true, 0, false, 1, true, true, false, true, 2><<<grid, 256, 0, s>>>(g);
        else if (D == 64 && g_flash_sm == 4)
            k_flash_attn<64, 2, true, 0, false, 1, true, true, false, true, 4><<<grid, 256, 0, s>>>(g);
        else if (D == 64 && g_flash_sm == 6)
            k_flash_attn<64, 2, true, 0, false, 1, true, true, false, true, 6><<<grid, 256, 0, s>>>(g);
        else if (D == 64)
            k_flash_attn<64, 2, true, 0, false, 1, true, true, false, true><<<grid, 256, 0, s>>>(g);
        else if (g_flash_pk)
            k_flash_attn<128, 4, true, 0, false, 1, true, true, false, true, 1><<<grid, 256, 0, s>>>(g);
        else if (g_flash_sm & 4)
            k_flash_attn<128, 4, true, 0, false, 1, true, true, false, true, 4><<<grid, 256, 0, s>>>(g);
        else
            k_flash_attn<128, 4, true, 0, false, 1, true, true, false, true><<<grid, 256, 0, s>>>(g);
        return;
    }
    if (qb2 && (g_flash_vpf & 1) && (g_flash_vtr & 1) && g_flash_ovl >= 2) {
        k_flash_attn<48, 2, true, 0, false, 2, true, true, true><<<grid, 256, 0, s>>>(g);
        return;
    }
    if (qb2 && (g_flash_vpf & 1) && (g_flash_vtr & 1)) {
        if (D == 40 && g_flash_mslot)
            k_flash_attn<48, 2, true, 0, true, 2, true, true><<<grid, 256, 0, s>>>(g);
        else
            k_flash_attn<48, 2, true, 0, false, 2, true, true><<<grid, 256, 0, s>>>(g);
        return;
    }
    if (qb2 && (g_flash_vpf & 1)) {
        if (D == 40 && g_flash_mslot)
            k_flash_attn<48, 2, true, 0, true, 2, true><<<grid, 256, 0, s>>>(g);
        else
            k_flash_attn<48, 2, true, 0, false, 2, true><<<grid, 256, 0, s>>>(g);
        return;
    }
    if (qb2) {
        if (D == 40 && g_flash_mslot)
            k_flash_attn<48, 2, true, 0, true, 2><<<grid, 256, 0, s>>>(g);
        else if (D <= 48)
            k_flash_attn<48, 2, true, 0, false, 2><<<grid, 256, 0, s>>>(g);
        else
            k_flash_attn<64, 2, true, 0, false, 2><<<grid, 256, 0, s>>>(g);
        return;
    }
    if (D == 40 && fast && g_flash_mslot && (g_flash_vpf & 1) && (g_flash_vtr & 1))
        k_flash_attn<48, 2, true, 0, true, 1, true, true><<<grid, 256, 0, s>>>(g);
    else if (D == 40 && fast && g_flash_mslot && (g_flash_vpf & 1))
        k_flash_attn<48, 2, true, 0, true, 1, true><<<grid, 256, 0, s>>>(g);
    else if (D == 40 && fast && g_flash_mslot)
        k_flash_attn<48, 2, true, 0, true><<<grid, 256, 0, s>>>(g);
    else if (D <= 48)
        FA_CASE(48, 2);
    else if (D <= 64)
        FA_CASE(64, 2);
    else if (D <= 80)
        FA_CASE(80, 3);
    else if (D <= 96)
        FA_CASE(96, 3);
    else if (D <= 128)
        FA_CASE(128, 4);
    else
        FA_CASE(160, 5);  // one workgroup per CU (80 accumulator + 40 Q + 40 prefetch registers; 86 KB of tiles)
#undef FA_CASE
}

}  // namespace mi355x
