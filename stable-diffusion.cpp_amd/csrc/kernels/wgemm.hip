// wgemm.hip — one-time weight re-layout for the MFMA kernels of gemm16.hip: static weights (immutable once uploaded) are decoded
// from their file type (f32 / f16 / bf16 / q8_0 / q4_0) into f16 MFMA fragment order
// [col/32][Kp/16][64 lanes][8 halfs], so a wave's B fragment is one contiguous 1 KiB piece for the LDS-DMA engine.
// (The first-generation k_linear_mfma / k_conv2d_mfma kernels that used to live here — 3 % of the MFMA peak — were removed in round 2;
// quantised Linear weights with few token rows are NOT expanded: they go through the in-register dequant kernels of qgemm.hip.)
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
// geglu_inner < 0: the 16-column interleave of -geglu_inner outputs (below);
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
    } else if (geglu_inner < 0) {
        // 16-column interleave (gemm16.hip epi_geglu16, any tile geometry): 32-row block rb of the image holds the value rows of outputs
        // 16 rb .. 16 rb + 15 in its rows 0..15 and their gate rows in rows 16..31
        const int64_t inner = -geglu_inner, out = rb * 16 + (lane & 15);
        row = out < inner ? ((lane & 16) ? inner : 0) + out : R;
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
                            int64_t total, int kblk) {
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
        if (kblk > 0) {  // gemm16 / conv3w: k = (channel block * taps + tap) * kblk + ic % kblk
            const int64_t taps = (int64_t)KH * KW, kb = k / kblk;
            tap = kb % taps;
            ic  = (kb / taps) * kblk + k % kblk;
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
void launch_wswz_conv(hipStream_t s, void* dst, const void* src, int64_t KW, int64_t KH, int64_t IC, int64_t OC, int kblk) {
    const int64_t ICp = rup(IC, 64), Kp = ICp * KW * KH, Rp = rup(OC, 128);
    const int64_t total = (Rp / 32) * (Kp / 16) * 64;
    k_wswz_conv<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((half8_t*)dst, (const __half*)src, (int)KW, (int)KH, IC, OC, ICp, Kp, total, kblk);
}

}  // namespace mi355x
