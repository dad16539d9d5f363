// gemm_generic.hip — ggml MUL_MAT for ANY operand pair (activations x activations, un-swizzled weights,
// quantised blocks) on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TF peak).
// This is the always-correct path; the static-weight hot GEMMs / convs go through wgemm.hip instead.
// Also: IM2COL (F16/F32 dst) for the unfused conv path.
#include "device_utils.h"
#include "kernels.h"
#include "ktime.h"

namespace mi355x {

enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q8_0 = 8, T_BF16 = 30 };

// load 8 consecutive k of one row (k0 multiple of 8) as f32, with in-register dequant for q8_0 / q4_0 blocks
__device__ __forceinline__ void load8(const char* row, int type, int64_t k0, int64_t K, float out[8]) {
    if (type == T_F32) {
        const float* p = (const float*)row + k0;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (k0 + j < K) ? p[j] : 0.f;
    } else if (type == T_F16) {
        const __half* p = (const __half*)row + k0;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (k0 + j < K) ? __half2float(p[j]) : 0.f;
    } else if (type == T_BF16) {
        const uint16_t* p = (const uint16_t*)row + k0;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (k0 + j < K) ? __uint_as_float((uint32_t)p[j] << 16) : 0.f;
    } else if (type == T_Q8_0) {  // { half d; int8 qs[32] } = 34 B
        if (k0 >= K) {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = 0.f;
            return;
        }
        const char* blk = row + (k0 / 32) * 34;
        const float d   = __half2float(*(const __half*)blk);
        const int8_t* q = (const int8_t*)(blk + 2) + (k0 & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = d * (float)q[j];
    } else {  // Q4_0: { half d; uint8 qs[16] } = 18 B; elem j<16 low nibble of qs[j]; j>=16 high nibble of qs[j-16]
        if (k0 >= K) {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = 0.f;
            return;
        }
        const char* blk  = row + (k0 / 32) * 18;
        const float d    = __half2float(*(const __half*)blk);
        const uint8_t* q = (const uint8_t*)(blk + 2);
        const int e0     = (int)(k0 & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + j;
            const int v = e < 16 ? (q[e] & 0xF) : (q[e - 16] >> 4);
            out[j]      = d * (float)(v - 8);
        }
    }
}

struct MMArgs {
    int64_t K, M, N;
    int64_t a_nb1, a_nb2, a_nb3, b_nb1, b_nb2, b_nb3, d_nb1, d_nb2, d_nb3;
    int64_t ne12, r2, r3;
    int a_type, b_type;
};

// D[i=n][j=m]: A-operand rows come from src1 (n), B-operand columns from src0 (m) -> lanes run along m (contiguous in dst)
__global__ __launch_bounds__(256) void k_mul_mat_generic(float* __restrict__ dst, const char* __restrict__ a, const char* __restrict__ b, MMArgs g) {
    __shared__ float As[64][33];  // src0 tile  [m][k]
    __shared__ float Bs[64][33];  // src1 tile  [n][k]
    const int64_t batch = blockIdx.z;
    const int64_t i12 = batch % g.ne12, i13 = batch / g.ne12;
    const int64_t i02 = i12 / g.r2, i03 = i13 / g.r3;
    a += i02 * g.a_nb2 + i03 * g.a_nb3;
    b += i12 * g.b_nb2 + i13 * g.b_nb3;
    dst = (float*)((char*)dst + i12 * g.d_nb2 + i13 * g.d_nb3);
    const int64_t m0 = (int64_t)blockIdx.x * 64, n0 = (int64_t)blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 2 waves, each a 32(m) x 32(n) block
    float16_t acc = {0};
    const int lrow = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 8;
    for (int64_t k0 = 0; k0 < g.K; k0 += 32) {
        float va[8], vb[8];
        if (m0 + lrow < g.M)
            load8(a + (m0 + lrow) * g.a_nb1, g.a_type, k0 + lk, g.K, va);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) va[j] = 0.f;
        if (n0 + lrow < g.N)
            load8(b + (n0 + lrow) * g.b_nb1, g.b_type, k0 + lk, g.K, vb);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) vb[j] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            As[lrow][lk + j] = va[j];
            Bs[lrow][lk + j] = vb[j];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float fa = Bs[wn * 32 + (lane & 31)][kk + (lane >> 5)];  // A operand: rows i = n
            const float fb = As[wm * 32 + (lane & 31)][kk + (lane >> 5)];  // B operand: cols j = m
            acc            = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
    }
    const int64_t m = m0 + wm * 32 + (lane & 31);
    if (m < g.M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < g.N) *(float*)((char*)dst + n * g.d_nb1 + m * 4) = acc[r];
        }
    }
}

void launch_mul_mat_generic(hipStream_t s, float* dst, const int64_t dne[4], const int64_t dnb[4], const View4& a, const View4& b) {
    KScope ks_(s, KF_GEMM_F32, 2.0 * (double)dne[0] * dne[1] * dne[2] * dne[3] * (double)a.ne[0], 0.0);
    MMArgs g;
    g.K = a.ne[0];
    g.M = a.ne[1];
    g.N = b.ne[1];
    g.a_nb1 = a.nb[1];
    g.a_nb2 = a.nb[2];
    g.a_nb3 = a.nb[3];
    g.b_nb1 = b.nb[1];
    g.b_nb2 = b.nb[2];
    g.b_nb3 = b.nb[3];
    g.d_nb1 = dnb[1];
    g.d_nb2 = dnb[2];
    g.d_nb3 = dnb[3];
    g.ne12  = b.ne[2];
    g.r2    = b.ne[2] / a.ne[2];
    g.r3    = b.ne[3] / a.ne[3];
    g.a_type = a.type;
    g.b_type = b.type;
    dim3 grid((unsigned)((g.M + 63) / 64), (unsigned)((g.N + 63) / 64), (unsigned)(b.ne[2] * b.ne[3]));
    k_mul_mat_generic<<<grid, 256, 0, s>>>(dst, (const char*)a.data, (const char*)b.data, g);
}

// ---------------------------------------------------------------------------------------- im2col
// dst [IC*KH*KW, OW, OH, N] (F16 or F32); K order (ic, kh, kw), kw fastest
struct I2CArgs {
    int64_t IW, IH, IC, N, KW, KH, OW, OH;
    int64_t x_nb0, x_nb1, x_nb2, x_nb3;
    int s0, s1, p0, p1, d0, d1;
    int dst_f16;
};
__global__ void k_im2col(char* __restrict__ dst, const char* __restrict__ x, I2CArgs g, int64_t total) {
    const int64_t CK = g.IC * g.KH * g.KW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k   = i % CK;
        const int64_t pos = i / CK;
        const int64_t ow = pos % g.OW, oh = (pos / g.OW) % g.OH, n = pos / (g.OW * g.OH);
        const int64_t kw = k % g.KW, kh = (k / g.KW) % g.KH, ic = k / (g.KW * g.KH);
        const int64_t iw = ow * g.s0 + kw * g.d0 - g.p0, ih = oh * g.s1 + kh * g.d1 - g.p1;
        float v = 0.f;
        if (iw >= 0 && iw < g.IW && ih >= 0 && ih < g.IH) v = *(const float*)(x + iw * g.x_nb0 + ih * g.x_nb1 + ic * g.x_nb2 + n * g.x_nb3);
        if (g.dst_f16)
            ((__half*)dst)[i] = __float2half_rn(v);
        else
            ((float*)dst)[i] = v;
    }
}
void launch_im2col_f16(hipStream_t s, void* dst, int dst_type, const View4& x, int64_t KW, int64_t KH, int64_t OW, int64_t OH, int s0, int s1, int p0,
                       int p1, int d0, int d1) {
    I2CArgs g;
    g.IW = x.ne[0];
    g.IH = x.ne[1];
    g.IC = x.ne[2];
    g.N  = x.ne[3];
    g.KW = KW;
    g.KH = KH;
    g.OW = OW;
    g.OH = OH;
    g.x_nb0 = x.nb[0];
    g.x_nb1 = x.nb[1];
    g.x_nb2 = x.nb[2];
    g.x_nb3 = x.nb[3];
    g.s0 = s0;
    g.s1 = s1;
    g.p0 = p0;
    g.p1 = p1;
    g.d0 = d0;
    g.d1 = d1;
    g.dst_f16 = dst_type == T_F16;
    const int64_t total = g.IC * KH * KW * OW * OH * g.N;
    int64_t blocks      = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    k_im2col<<<(unsigned)blocks, 256, 0, s>>>((char*)dst, (const char*)x.data, g, total);
}

}  // namespace mi355x
