// conv3w.hip — 3x3 / stride-1 implicit-GEMM conv for gfx950 with the INPUT WINDOW RESIDENT IN LDS (round 3).
//
// Why: the round-2 kernel (k_gemm16<256,320,true,...,PIPE>) fetched an A tile per (channel block, tap) through the LDS-DMA engine and computed
// the per-lane tap source addresses (mask test, zero page select, 64-bit adds, a divergent branch) in front of EVERY DMA piece — ~160 non-MFMA
// instructions per 20 MFMAs per wave.  With two waves per SIMD the loop was bound by instruction ISSUE, not by the matrix pipe or the fabric
// (profiles/r02h_t320_ablation.txt: 1384 ns per 32-wide k stage against 610 ns of MFMA work; no-DMA variant 1187 ns; the steady-state loop in the
// disassembly: 20 v_mfma, 14 ds_read_b128, 5 LDS-DMA and 140 scalar / vector address instructions).  The nine taps of a 3x3 conv read the SAME
// input pixels shifted by (kh, kw): this kernel stages the (TR+2) x (TW+2) pixel window of a 32-channel block ONCE (LDS-DMA, double-buffered
// across channel blocks) and every tap's A fragment is a ds_read_b128 at window_address + IMMEDIATE(kh) from one of three per-kw address
// registers.  What is left per 32-wide k stage: 20 MFMAs, 14 fragment reads, 2-3 weight DMA pieces with pointer bumps, one barrier.
// A-side L2 -> LDS traffic drops 6x (window once instead of 9 tap tiles), so the fabric port sees ~1.5x the NHWC image instead of ~3x.
//
// Geometry: workgroup = 256 consecutive raster positions (TR = 256 / TW full rows of a TW-wide image) x BN output channels; 8 waves = 4 row
// groups (64 positions) x 2 column halves, each wave 2 x (BN/64) accumulator blocks of 32x32 (v_mfma_f32_32x32x16_f16, D[oc][pos] like the
// round-2 conv, so epi_conv is shared).  K order = (32-channel block, tap, channel): one stage = one tap of one channel block = 2 MFMA k-steps.
// LDS: 2 window buffers (pixel = 64 B = 4 16-byte k-slots, row pitch TW+16 pixels so that the XOR swizzle (slot ^ ((x >> 2) & 3)) does not
// depend on the row: conflict-free ds_read_b128 for any tap shift) + a 4-stage ring of weight tiles (BN x 32 k in MFMA fragment order).
// Pipeline = the round-2 PIPE loop: fragments of the next k-step are read while the current one multiplies; one barrier per stage at the head
// of its second k-step (publishes stage g+1, frees stage g's slot for the DMA of stage g+4); counted vmcnt.
#include <cstdio>
#include <cstdlib>

#include "device_utils.h"
#include "g16_common.h"
#include "kernels.h"
#include "ktime.h"

namespace mi355x {

template <int TW, int BN>
struct C3Geom {
    static constexpr int TR     = 256 / TW;            // image rows per tile
    static constexpr int PITCH  = TW + 16;             // window row pitch (pixels), a multiple of 16
    static constexpr int WROWS  = TR + 2;
    static constexpr int WBYTES = WROWS * PITCH * 64;  // one 32-channel window
    static constexpr int NWP    = WBYTES / 1024;       // 1-KiB DMA pieces (16 pixels each)
    static constexpr int NWPW   = (NWP + 7) / 8;       // pieces per wave (the last few of the 8 * NWPW are dummies fetched from the zero page)
    static constexpr int WBUF   = NWPW * 8 * 1024;
    static constexpr int CB     = BN / 64;             // 32-column blocks per wave (2 wave columns)
    static constexpr int NF     = (BN / 32) * 2;       // weight fragments per stage
    static constexpr int BSTAGE = NF * 1024;
    static constexpr int NST    = 4;
    static constexpr int LDS    = 2 * WBUF + NST * BSTAGE;
    // byte distance of the wave's second 32-position block inside the window: 32 pixels on (TW >= 64), one row down (TW = 32), two rows down (TW = 16)
    static constexpr int RB1    = TW >= 64 ? 32 * 64 : (32 / TW) * PITCH * 64;
    static_assert(WBYTES % 1024 == 0 && PITCH % 16 == 0 && NWPW <= 5 && LDS <= 160 * 1024, "window geometry");
};

#define C3_RD(DST_, ADDR_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_))
#define C3_TIE(X_) asm volatile("" : "+v"(X_))

// PRIO: wave priority around the MFMA groups — 0: none, 1: s_setprio 1 around every MFMA pair, 2: s_setprio 1 from the first MFMA of a k-step to its
// last, 3 (default): static, the second-dispatched half of the workgroup (waves 4..7) runs at priority 1 (MI355X_MICROARCH.md, two waves per SIMD,
// item 4).  Measured on the SD1.5 step (profiles/r05d_ab_conv3w_prio.txt): 119.3 / 120.0 / 119.5 / 118.6 us per launch for 0 / 1 / 2 / 3 — the loop is not
// bound by which of a SIMD's two waves issues first; the static form is kept for its 0.6 %.  Option "conv3w_prio" = 0 selects the plain kernel (A/B).
template <int TW, int BN, int PRIO = 3>
__global__ __launch_bounds__(512, 2) void k_conv3w(G16Args g) {
    using G = C3Geom<TW, BN>;
    constexpr int CB = G::CB, CL = CB - 2, PITCH = G::PITCH, NWPW = G::NWPW, WBUF = G::WBUF, BSTAGE = G::BSTAGE, NF = G::NF, RB1 = G::RB1;
    static_assert(CB >= 3, "at least three column blocks per wave");
    __shared__ __attribute__((aligned(1024))) char smem[G::LDS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave & 3, wc = wave >> 2;
    const int hi = lane >> 5;
    if constexpr (PRIO == 3) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }

    int bid, kslice;
    g16_wg_order(g, g.ncol_tiles, bid, kslice);  // default: XCD-aware, consecutive tiles of one XCD share rows / halos; weight-heavy launches: weight-major
    const int row_tile = bid / g.ncol_tiles;
    const int col_tile = bid - row_tile * g.ncol_tiles;
    const int64_t row0 = (int64_t)row_tile * 256;
    const int col0     = col_tile * BN;

    // K range of this workgroup: 32-channel blocks [icb0, icb0 + nicb)
    int icb0 = 0, nicb = g.nt;  // g.nt = number of 32-channel blocks for this kernel
    if (g.split_k > 1) {
        icb0 = kslice * g.nt_slice;
        nicb = min(g.nt_slice, g.nt - icb0);
        g.dst += (int64_t)kslice * g.slab;
    }

    // ---- tile position: image n, first row oy0 (TW == image width: a tile is TR full rows)
    const int img = (int)(row0 / g.OHOW);
    const int oy0 = (int)(row0 - (int64_t)img * g.OHOW) / TW;

    // ---- window DMA sources: wave w stages pieces w * NWPW + j; piece p = window pixels [16p, 16p + 16), lane i -> pixel i / 4, physical k-slot i % 4
    // holding logical slot (i & 3) ^ ((i >> 4) & 3)  (= slot ^ ((x >> 2) & 3): the pitch is a multiple of 16 pixels)
    uint32_t wofsA[NWPW];  // byte offset of this lane's 16 bytes: from g.A (real pixels) or from the zero page (padding / dummy lanes)
    uint32_t wokA = 0;     // bit j: piece j of this lane reads a real pixel
#pragma unroll
    for (int j = 0; j < NWPW; ++j) {
        const int p   = wave * NWPW + j;
        const int pl  = p * 16 + (lane >> 2);           // linear window pixel
        const int wy  = pl / PITCH, wx = pl - wy * PITCH;
        const int iy  = oy0 - 1 + wy, ix = wx - 1;
        const int ls  = (lane & 3) ^ ((lane >> 4) & 3);
        const bool ok = p < G::NWP && wx < TW + 2 && iy >= 0 && iy < g.H && ix >= 0 && ix < TW;
        wofsA[j]      = ok ? (uint32_t)(((((int64_t)img * g.H + iy) * TW + ix) * g.ICp + ls * 8) * 2) : (uint32_t)(ls * 16);
        wokA |= ok ? (1u << j) : 0u;
    }
    const char* abase = (const char*)g.A + (int64_t)icb0 * 64;  // wave-uniform: the channel block being staged (advances by 64 bytes per block)
    // ---- weight DMA sources: stage s (= one tap of one channel block) holds fragments f = cb * 2 + ks, f < NF; wave w fetches f = w + 8q
    constexpr int NPBMAX = (NF + 7) / 8;
    const half8_t* wsrcB[NPBMAX];
    int wdstB[NPBMAX];
#pragma unroll
    for (int q = 0; q < NPBMAX; ++q) {
        const int f  = q * 8 + wave;
        const int fs = f < NF ? f : NF - 1;
        const int cb = fs >> 1, ks = fs & 1;
        wsrcB[q]     = g.W + ((int64_t)(col0 / 32 + cb) * g.kfr + (int64_t)icb0 * 18 + ks) * 64 + lane;
        wdstB[q]     = fs * 1024;
    }
    const bool w_short = (NF % 8) != 0 && wave >= (NF % 8);  // this wave fetches one weight fragment fewer per stage

    float16_t acc[2][CB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) acc[a][b] = (float16_t){0};

    // ---- fragment read addresses.  A: one register per (kw, k-step): window pixel (oyl + kh, xb + (lane & 31) + kw), kh and the second row block are
    // immediates; B: lane-linear fragments of the current ring slot
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // this lane's output position inside the window (before the tap shift): a 32-position MFMA block is half a row (TW >= 64), a row (32) or two rows (16)
    const int oyl = (wr * 64) / TW + (TW < 32 ? (lane & 31) / TW : 0);
    const int xb  = TW >= 64 ? (wr * 64) % TW + (lane & 31) : (lane & 31) % TW;
    uint32_t aw[3];  // k-step 0; k-step 1 reads the slot with bit 1 flipped: address ^ 32 (the window base is 1-KiB aligned)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int x = xb + kw;
        aw[kw]      = lds0 + (uint32_t)((oyl * PITCH + x) * 64 + ((hi ^ ((x >> 2) & 3)) << 4));
    }
    const uint32_t bad0 = lds0 + (uint32_t)(2 * WBUF + wc * CB * 2 * 1024 + lane * 16);

    auto mma = [&](int rb, int cb, const half8_t& a, const half8_t& b) { acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[rb][cb], 0, 0, 0); };  // D[oc][pos]

    half8_t A0[2], A1[2], BL[CL], BH0[2], BH1[2];
    int slot = 0;          // ring slot of the current stage
    int wb   = 0;          // window buffer of the current channel block
    uint32_t bcur = bad0;  // B fragment base of the current stage's slot

// one LDS-DMA piece of the weight stage that goes into ring slot FSLOT_ (Q_-th fragment of this wave), then the pointer moves on one stage; one piece
// of the NEXT channel block's window into buffer NWB_.  (Macros on locals, not lambdas: a mutating by-reference capture made the compiler keep
// such state in scratch memory in gemm16.hip.)
#define C3_DMA_B(Q_, FSLOT_)                                                      \
    do {                                                                          \
        GLDS16(wsrcB[Q_], smem + 2 * WBUF + (FSLOT_) * BSTAGE + wdstB[Q_]);       \
        wsrcB[Q_] += 2 * 64; /* next stage: two k-steps of 64 lanes */            \
    } while (0)
#define C3_DMA_W(J_, NWB_)                                                                                          \
    do {                                                                                                            \
        const char* p_ = (((wokA >> (J_)) & 1u) ? abase : (const char*)g.zero) + wofsA[J_];                         \
        GLDS16(p_, smem + (NWB_) * WBUF + (wave * NWPW + (J_)) * 1024);                                             \
    } while (0)
// One k-step (cf. G16_KSTEP_H in gemm16.hip).  ACUR / ANXT, HCUR / HNXT: current and next fragment sets; an_ + AO_: LDS address of this lane's A
// row (row block 0) for the NEXT k-step; bn_: first B fragment of this wave for the NEXT k-step, KSN_ its k-step index inside its stage.
// HOOK_(i), i = 0 .. 3: a statement issued behind MFMA group i.
#define C3_KSTEP(ACUR, ANXT, HCUR, HNXT, an_, AO_, bn_, KSN_, HOOK_, ...)                                            \
    do {                                                                                                             \
        C3_RD(ANXT[0], an_, (AO_));                                                                                  \
        C3_RD(ANXT[1], an_, (AO_) + RB1);                                                                            \
        C3_RD(HNXT[0], bn_, ((CB - 2) * 2 + (KSN_)) * 1024);                                                         \
        C3_RD(HNXT[1], bn_, ((CB - 1) * 2 + (KSN_)) * 1024);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C3_PRIO_UP(1); C3_PRIO_UP(2);                                                                                \
        mma(0, 0, ACUR[0], BL[0]);                                                                                   \
        mma(1, 0, ACUR[1], BL[0]);                                                                                   \
        C3_PRIO_DN(1);                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C3_RD(BL[0], bn_, (0 * 2 + (KSN_)) * 1024);                                                                  \
        HOOK_(0, __VA_ARGS__);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C3_PRIO_UP(1);                                                                                               \
        mma(0, 1, ACUR[0], BL[1]);                                                                                   \
        mma(1, 1, ACUR[1], BL[1]);                                                                                   \
        C3_PRIO_DN(1);                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C3_RD(BL[1], bn_, (1 * 2 + (KSN_)) * 1024);                                                                  \
        HOOK_(1, __VA_ARGS__);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if constexpr (CL > 2) {                                                                                      \
            C3_PRIO_UP(1);                                                                                           \
            mma(0, 2, ACUR[0], BL[CL - 1]);                                                                          \
            mma(1, 2, ACUR[1], BL[CL - 1]);                                                                          \
            C3_PRIO_DN(1);                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            C3_RD(BL[CL - 1], bn_, (2 * 2 + (KSN_)) * 1024);                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        HOOK_(2, __VA_ARGS__);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C3_PRIO_UP(1);                                                                                               \
        mma(0, CB - 2, ACUR[0], HCUR[0]);                                                                            \
        mma(1, CB - 2, ACUR[1], HCUR[0]);                                                                            \
        C3_PRIO_DN(1);                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        HOOK_(3, __VA_ARGS__);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C3_PRIO_UP(1);                                                                                               \
        mma(0, CB - 1, ACUR[0], HCUR[1]);                                                                            \
        mma(1, CB - 1, ACUR[1], HCUR[1]);                                                                            \
        C3_PRIO_DN(1); C3_PRIO_DN(2);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#define C3_PRIO_UP(P_) do { if constexpr (PRIO == (P_)) __builtin_amdgcn_s_setprio(1); } while (0)
#define C3_PRIO_DN(P_) do { if constexpr (PRIO == (P_)) __builtin_amdgcn_s_setprio(0); } while (0)
#define C3_TIE_FRAGS(AS_, HS_)                                                                                       \
    do {                                                                                                             \
        C3_TIE(AS_[0]);                                                                                              \
        C3_TIE(AS_[1]);                                                                                              \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CL; ++cb_) C3_TIE(BL[cb_]);                                        \
        C3_TIE(HS_[0]);                                                                                              \
        C3_TIE(HS_[1]);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#define C3_NOHOOK(I_, ...) ((void)0)
// DMA hook of stage tap T_: weight stage g + 4 into the slot this stage just left (NP_ pieces, hook positions 0 .. NP_-1; a last block has nothing
// left to fetch from tap 5 on), then — taps < NWPW of a non-last block — one piece of the next block's window (position 3)
#define C3_DMAHOOK(I_, T_, NP_, LAST_, FSLOT_)                                                        \
    do {                                                                                              \
        if constexpr ((!(LAST_) || (T_) <= 4) && (I_) < (NP_)) C3_DMA_B(I_, FSLOT_);                  \
        if constexpr (!(LAST_) && (T_) < NWPW && (I_) == 3) C3_DMA_W(T_, wb ^ 1);                     \
    } while (0)

    // One stage = tap T_ (kh = T_ / 3, kw = T_ % 3) of the current channel block.  NP_ = weight pieces this wave issues per stage, LAST_ = last
    // channel block of this workgroup.  The DMA hook issues weight stage g + 4 into the slot this stage just left and, for taps < NWPW of a
    // non-last block, one piece of the next block's window.  VM_ = DMAs of this wave that may stay in flight across this stage's wait: everything
    // issued after weight stage g + 1 (see the derivation in DESIGN.md section 3.1): the window pieces of the hooks of stages g-3 .. g-1 and the
    // weight stages g + 2, g + 3 where they exist.
#define C3_W(T_) ((T_) >= 0 && (T_) < NWPW ? 1 : 0)
#define C3_STAGE(T_, NP_, LAST_)                                                                                                         \
    do {                                                                                                                                 \
        constexpr int kh_ = (T_) / 3, kw_ = (T_) % 3, tn_ = ((T_) + 1) % 9, khn_ = tn_ / 3, kwn_ = tn_ % 3;                             \
        if constexpr ((T_) == 0 && !(LAST_)) abase += 64; /* the window pieces issued during this block belong to the next one */        \
        constexpr int vm_ = (LAST_) ? (NP_) * (((T_) <= 6 ? 1 : 0) + ((T_) <= 5 ? 1 : 0))                                                \
                                    : 2 * (NP_) + C3_W((T_) - 3) + C3_W((T_) - 2) + C3_W((T_) - 1);                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                               \
        C3_TIE_FRAGS(A0, BH0);                                                                                                           \
        const uint32_t a1_ = aw[kw_] ^ 32u;                                                                                              \
        C3_KSTEP(A0, A1, BH0, BH1, a1_, kh_ * PITCH * 64, bcur, 1, C3_NOHOOK, 0);                                                        \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(vm_) : "memory");                                                            \
        C3_TIE_FRAGS(A1, BH1);                                                                                                           \
        asm volatile("s_barrier" ::: "memory");                                                                                          \
        const int fslot_ = slot;                                                                                                         \
        slot             = (slot + 1) & 3;                                                                                               \
        bcur             = bad0 + (uint32_t)slot * BSTAGE;                                                                               \
        if constexpr ((T_) == 8 && !(LAST_)) { /* the remaining reads of this stage fetch the next block's window */                    \
            wb ^= 1;                                                                                                                     \
            const uint32_t d_ = wb ? (uint32_t)WBUF : (uint32_t)(0 - WBUF);                                                              \
            _Pragma("unroll") for (int a_ = 0; a_ < 3; ++a_) aw[a_] += d_;                                                               \
        }                                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                               \
        C3_KSTEP(A1, A0, BH1, BH0, aw[kwn_], khn_ * PITCH * 64, bcur, 0, C3_DMAHOOK, T_, NP_, LAST_, fslot_);                            \
    } while (0)
#define C3_BLOCK(NP_, LAST_)          \
    do {                              \
        C3_STAGE(0, NP_, LAST_);      \
        C3_STAGE(1, NP_, LAST_);      \
        C3_STAGE(2, NP_, LAST_);      \
        C3_STAGE(3, NP_, LAST_);      \
        C3_STAGE(4, NP_, LAST_);      \
        C3_STAGE(5, NP_, LAST_);      \
        C3_STAGE(6, NP_, LAST_);      \
        C3_STAGE(7, NP_, LAST_);      \
        C3_STAGE(8, NP_, LAST_);      \
    } while (0)
#define C3_MAIN(NP_)                                                                                                         \
    do {                                                                                                                     \
        /* prologue: window of the first block, weight stages 0 .. 3; stage 0 + the window landed -> barrier -> first reads */ \
        _Pragma("unroll") for (int j_ = 0; j_ < NWPW; ++j_) C3_DMA_W(j_, 0);                                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                                   \
            if constexpr ((NP_) > 0) C3_DMA_B(0, s_);                                                                        \
            if constexpr ((NP_) > 1) C3_DMA_B(1, s_);                                                                        \
            if constexpr ((NP_) > 2) C3_DMA_B(2, s_);                                                                        \
        }                                                                                                                    \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NP_)) : "memory");                                                     \
        asm volatile("s_barrier" ::: "memory");                                                                              \
        C3_RD(A0[0], aw[0], 0);                                                                                              \
        C3_RD(A0[1], aw[0], RB1);                                                                                            \
        C3_RD(BL[0], bcur, (0 * 2) * 1024);                                                                                  \
        C3_RD(BL[1], bcur, (1 * 2) * 1024);                                                                                  \
        if constexpr (CL > 2) C3_RD(BL[CL - 1], bcur, (2 * 2) * 1024);                                                       \
        C3_RD(BH0[0], bcur, ((CB - 2) * 2) * 1024);                                                                          \
        C3_RD(BH0[1], bcur, ((CB - 1) * 2) * 1024);                                                                          \
        for (int ib_ = 0; ib_ + 1 < nicb; ++ib_) C3_BLOCK(NP_, false);                                                       \
        C3_BLOCK(NP_, true);                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    } while (0)

    if constexpr (NF % 8 != 0) {
        if (w_short)
            C3_MAIN(NPBMAX - 1);
        else
            C3_MAIN(NPBMAX);
    } else {
        C3_MAIN(NPBMAX);
    }

    // ---- epilogue (shared with the round-2 conv): NCHW f32 scatter + bias (+ per-(image, channel) embedding add) (+ residual)
    if (g.ep.residual)
        epi_conv<1>(acc, g, row0, col0, wr, wc, lane);
    else
        epi_conv<0>(acc, g, row0, col0, wr, wc, lane);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static int g_conv3w_prio = 3;  // option "conv3w_prio": 0 = the kernel without the static wave priority (320-column tiles on 16 / 32 / 64-wide maps; A/B runs)
void conv3w_set_prio(int v) { g_conv3w_prio = v; }
static int g_conv3w = 1;  // option "conv3w": 0 = every conv on the round-2 per-tap gather kernel (A/B measurements)
void conv3w_set(int v) { g_conv3w = v; }
static int g_conv3w_min_blocks = 8, g_conv3w_min_blocks_deep = 5;  // options "conv3w_min_blocks" / "conv3w_min_blocks_deep": 32-channel blocks (9 stages each) a K slice keeps
void conv3w_set_min_blocks(int v) { g_conv3w_min_blocks = v > 0 ? v : 1; }
void conv3w_set_min_blocks_deep(int v) { g_conv3w_min_blocks_deep = v > 0 ? v : 1; }

// Shapes the window kernel takes: 3x3, stride 1, no fused upsample, image width 16 / 32 / 64 / 128 with whole tiles of 256 positions per image,
// OC a multiple of the column tile (320 or 256), a launch that fills the chip (with K slices over the 32-channel blocks where the output
// alone does not).  Returns the number of K slices (>= 1), or 0 when the shape stays on the round-2 kernel.
int conv3w_plan(int64_t W, int64_t H, int64_t IC, int64_t N, int64_t OC, int ksize, int stride, bool upscale2x, int* bn_out) {
    if (!g_conv3w || ksize != 3 || stride != 1 || upscale2x) return 0;
    if (!(W == 16 || W == 32 || W == 64 || W == 128) || (H * W) % 256 != 0 || IC < 64) return 0;
    if (N * H * W * ((IC + 63) / 64 * 64) * 2 >= (1ll << 32)) return 0;  // the window DMA keeps 32-bit byte offsets into the NHWC image
    const int bn = OC % 320 == 0 ? 320 : (OC % 256 == 0 ? 256 : 0);
    if (!bn) return 0;
    const int64_t tiles = (H * W * N / 256) * (OC / bn);
    const int64_t nicb  = (IC + 63) / 64 * 2;
    int S = 1;
    if (tiles < 192) {
        // one workgroup per CU and too few tiles: K slices over the 32-channel blocks.  Every slice writes a whole f32 slab and a reduce pass reads
        // them all, so a slice must keep real work (measured against the round-2 kernel, profiles/r04b_conv3w_check.txt): >= 8 blocks (72 stages) per
        // slice wins 10-15 %; 5 blocks per slice lose 8-12 % at 128 tiles (320 -> 640 @32x32 x 16, 320 -> 320 @128x128 x 2) but win 15 % where
        // the round-2 kernel is weakest (<= 64 tiles x 4 slices: 640 -> 640 @64x64 x 2)
        S = (int)(256 / tiles);
        if (S > 8) S = 8;
        const int64_t minb = tiles <= 64 ? g_conv3w_min_blocks_deep : g_conv3w_min_blocks;
        if (tiles <= 64 && S > 4) S = 4;
        while (S > 1 && (nicb / S < minb || (S - 1) * ((nicb + S - 1) / S) >= nicb)) --S;  // every slice keeps a minimum of blocks, none is empty
        if (tiles * S < 128) return 0;
    } else {
        const int64_t rounds = (tiles + 255) / 256;
        if (tiles * 4 < rounds * 256 * 3) return 0;  // fills < 75 % of its rounds: the finer round-2 tiles quantise better
    }
    if (bn_out) *bn_out = bn;
    return S;
}

template <int TW, int BN>
static void c3_launch(hipStream_t s, const G16Args& g, unsigned tiles, unsigned ny) {
    if constexpr (BN == 320 && TW <= 64) {
        if (g_conv3w_prio == 0) return (void)k_conv3w<TW, BN, 0><<<dim3(tiles, ny), 512, 0, s>>>(g);
    }
    k_conv3w<TW, BN><<<dim3(tiles, ny), 512, 0, s>>>(g);
}

// zero page shared with gemm16.hip
const _Float16* gemm16_zero_page();
void launch_splitk_reduce_conv_gn(hipStream_t s, float* dst, const float* ws, int S, int64_t hw, int64_t C, int64_t N, const Epilogue& e);
void launch_splitk_reduce_conv(hipStream_t s, float* dst, const float* ws, int S, int64_t n, const float* bias, int64_t inner, int64_t C, const float* residual,
                               const float* chan_add, int64_t chan_ld);

void launch_conv3w(hipStream_t s, float* dst, const void* x16_nhwc, const void* wswz32, int64_t W, int64_t H, int64_t IC, int64_t N, int64_t OC, const Epilogue& e,
                   float* splitk_ws, int S) {
    int bn = 0;
    if (conv3w_plan(W, H, IC, N, OC, 3, 1, false, &bn) == 0) {
        fprintf(stderr, "ggml-mi355x: launch_conv3w called for a shape it does not take\n");
        abort();
    }
    if (e.act >= 0) {
        fprintf(stderr, "ggml-mi355x: conv3w has no fused activation\n");
        abort();
    }
    G16Args g{};
    g.A    = (const _Float16*)x16_nhwc;
    g.W    = (const half8_t*)wswz32;
    g.ICp  = (int)((IC + 63) / 64 * 64);
    g.kfr  = (int64_t)g.ICp * 9 / 16;
    g.dst  = dst;
    g.H    = (int)H;
    g.Wd   = (int)W;
    g.OW   = (int)W;
    g.OH   = (int)H;
    g.OHOW = W * H;
    g.S    = 1;
    g.pad  = 1;
    g.KS   = 3;
    g.nt   = g.ICp / 32;  // 32-channel blocks
    g.R    = g.OHOW * N;
    g.C    = OC;
    g.zero = gemm16_zero_page();
    g.ep   = G16Epi{e.bias, e.residual, e.scale};
    g.ep.chan_add = e.chan_add;
    g.ep.chan_ld  = (int)e.chan_ld;
    g.ncol_tiles  = (int)(OC / bn);
    if (S > 1 && splitk_ws) {
        g.split_k  = S;
        g.nt_slice = (g.nt + S - 1) / S;
        g.slab     = g.R * OC;
        g.dst      = splitk_ws;
        g.ep       = G16Epi{nullptr, nullptr, e.scale};
    } else {
        S = 1;
    }
    const unsigned tiles = (unsigned)((g.R / 256) * g.ncol_tiles);
    g.worder             = gemm16_worder_rows(g, tiles, (unsigned)S);  // 16x16-level 1280-channel convs: 29.5 .. 59 MB of weights against 10.5 MB of image
    const double bytes   = (double)N * H * W * g.ICp * 2.0 + (double)g.ICp * 9 * ((OC + 127) / 128 * 128) * 2.0 + (double)g.R * OC * 4.0 * (e.residual ? 2.0 : 1.0);
    {
        KScope ks_(s, KF_CONV_T256, 2.0 * g.R * IC * 9 * OC, bytes);
        if (bn == 320) {
            if (W == 16) c3_launch<16, 320>(s, g, tiles, (unsigned)S);
            else if (W == 32) c3_launch<32, 320>(s, g, tiles, (unsigned)S);
            else if (W == 64) c3_launch<64, 320>(s, g, tiles, (unsigned)S);
            else c3_launch<128, 320>(s, g, tiles, (unsigned)S);
        } else {
            if (W == 16) c3_launch<16, 256>(s, g, tiles, (unsigned)S);
            else if (W == 32) c3_launch<32, 256>(s, g, tiles, (unsigned)S);
            else if (W == 64) c3_launch<64, 256>(s, g, tiles, (unsigned)S);
            else c3_launch<128, 256>(s, g, tiles, (unsigned)S);
        }
    }
    if (S > 1) launch_splitk_reduce_conv_gn(s, dst, splitk_ws, S, g.OHOW, OC, N, e);
}

}  // namespace mi355x
