// norm.hip — GroupNorm(+affine+SiLU), LayerNorm / RMSNorm(+affine), row softmax for gfx950.
// HBM-bound: algorithmic bytes = 1 read + 1 write of the activation (the re-reads of the variance and
// write passes are served by L2 / Infinity Cache).  Two-pass mean / variance like ggml-cpu (SURVEY.md
// Appendix E.5: sums in double there; here f32 partials per lane + tree reduction, error ~1e-7 rel).
// Reductions: wavefront shuffles, then LDS across the waves of the block.
#include "device_utils.h"
#include "kernels.h"
#include "ktime.h"

namespace mi355x {

// one block per (image, group).  x/dst: [hw, C, N] contiguous f32.
template <int NT>
__global__ __launch_bounds__(NT) void k_group_norm(float* __restrict__ dst, const float* __restrict__ x, int64_t hw, int C, int groups, int cpg, float eps,
                                                   const float* __restrict__ w, const float* __restrict__ b, int silu) {
    __shared__ float scratch[2 * (NT / 64)];
    const int g = blockIdx.x % groups, n = blockIdx.x / groups;
    const int c0 = g * cpg, c1 = min(c0 + cpg, C);
    if (c0 >= c1) return;
    const int64_t cnt  = (int64_t)(c1 - c0) * hw;
    const int64_t base = ((int64_t)n * C + c0) * hw;
    const float* xs    = x + base;
    float* ys          = dst + base;
    const bool v4      = (hw % 4 == 0) && ((((uintptr_t)xs | (uintptr_t)ys) & 15) == 0);

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

    if (v4) {
        const int64_t hw4 = hw / 4;
        for (int64_t i = threadIdx.x; i < cnt / 4; i += NT) {
            const int c    = c0 + (int)(i / hw4);
            const float sc = w ? w[c] * rstd : rstd;
            const float sh = b ? b[c] : 0.f;
            float4 v       = ((const float4*)xs)[i];
            v.x            = (v.x - mean) * sc + sh;
            v.y            = (v.y - mean) * sc + sh;
            v.z            = (v.z - mean) * sc + sh;
            v.w            = (v.w - mean) * sc + sh;
            if (silu) {
                v.x = act_apply<UN_SILU>(v.x);
                v.y = act_apply<UN_SILU>(v.y);
                v.z = act_apply<UN_SILU>(v.z);
                v.w = act_apply<UN_SILU>(v.w);
            }
            ((float4*)ys)[i] = v;
        }
    } else {
        for (int64_t i = threadIdx.x; i < cnt; i += NT) {
            const int c    = c0 + (int)(i / hw);
            const float sc = w ? w[c] * rstd : rstd;
            const float sh = b ? b[c] : 0.f;
            float v        = (xs[i] - mean) * sc + sh;
            if (silu) v = act_apply<UN_SILU>(v);
            ys[i] = v;
        }
    }
}

void launch_group_norm(hipStream_t s, float* dst, const float* x, int64_t hw, int64_t C, int64_t N, int groups, float eps, const float* w,
                       const float* b, bool silu) {
    KScope ks_(s, KF_NORM_F32, 0.0, (double)hw * C * N * 8.0);
    const int cpg      = (int)((C + groups - 1) / groups);
    const int64_t cnt  = cpg * hw;
    const int blocks   = (int)(N * groups);
    if (cnt >= 16384)
        k_group_norm<1024><<<blocks, 1024, 0, s>>>(dst, x, hw, (int)C, groups, cpg, eps, w, b, silu ? 1 : 0);
    else
        k_group_norm<256><<<blocks, 256, 0, s>>>(dst, x, hw, (int)C, groups, cpg, eps, w, b, silu ? 1 : 0);
}

// one wave per row
// rows are addressed through RowMap: uniform stride, or a 3-level (i1, i2, i3) decomposition for strided views (the per-head q / k
// slices of a fused qkv projection, flux.hpp:283-296)
struct RowMap {
    int64_t ne1, ne2;          // 0: uniform
    int64_t s1, s2, s3;        // source strides (elements) of dims 1..3 (s1 = the uniform stride)
    int64_t d1, d2, d3;        // destination strides
};
__global__ __launch_bounds__(256) void k_layer_norm(float* __restrict__ dst, const float* __restrict__ x, int ne0, int64_t nrows, RowMap m, float eps,
                                                    const float* __restrict__ w, const float* __restrict__ b, int rms) {
    const int lane    = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float* xr;
    float* yr;
    if (m.ne1 == 0) {
        xr = x + row * m.s1;
        yr = dst + row * m.d1;
    } else {
        const int64_t i1 = row % m.ne1, t = row / m.ne1, i2 = t % m.ne2, i3 = t / m.ne2;
        xr = x + i1 * m.s1 + i2 * m.s2 + i3 * m.s3;
        yr = dst + i1 * m.d1 + i2 * m.d2 + i3 * m.d3;
    }
    const bool v4   = (ne0 % 4 == 0) && ((((uintptr_t)xr | (uintptr_t)yr) & 15) == 0) && (!w || (((uintptr_t)w & 15) == 0)) && (!b || (((uintptr_t)b & 15) == 0));
    float mean = 0.f, rstd;
    if (v4) {
        const int n4 = ne0 / 4;
        float s = 0.f;
        if (!rms) {
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
        rstd = rsqrtf(wave_sum(q) / (float)ne0 + eps);
        for (int i = lane; i < n4; i += 64) {
            float4 v = ((const float4*)xr)[i];
            v.x      = (v.x - mean) * rstd;
            v.y      = (v.y - mean) * rstd;
            v.z      = (v.z - mean) * rstd;
            v.w      = (v.w - mean) * rstd;
            if (w) {
                const float4 ww = ((const float4*)w)[i];
                v.x *= ww.x;
                v.y *= ww.y;
                v.z *= ww.z;
                v.w *= ww.w;
            }
            if (b) {
                const float4 bb = ((const float4*)b)[i];
                v.x += bb.x;
                v.y += bb.y;
                v.z += bb.z;
                v.w += bb.w;
            }
            ((float4*)yr)[i] = v;
        }
    } else {
        float s = 0.f;
        if (!rms) {
            for (int i = lane; i < ne0; i += 64) s += xr[i];
            mean = wave_sum(s) / (float)ne0;
        }
        float q = 0.f;
        for (int i = lane; i < ne0; i += 64) {
            const float a = xr[i] - mean;
            q += a * a;
        }
        rstd = rsqrtf(wave_sum(q) / (float)ne0 + eps);
        for (int i = lane; i < ne0; i += 64) {
            float v = (xr[i] - mean) * rstd;
            if (w) v *= w[i];
            if (b) v += b[i];
            yr[i] = v;
        }
    }
}

void launch_layer_norm(hipStream_t s, float* dst, const float* x, int64_t ne0, int64_t nrows, int64_t x_stride, int64_t d_stride, float eps,
                       const float* w, const float* b, bool rms) {
    KScope ks_(s, KF_NORM_F32, 0.0, (double)ne0 * nrows * 8.0);
    const int blocks = (int)((nrows + 3) / 4);
    RowMap m{0, 0, x_stride, 0, 0, d_stride, 0, 0};
    k_layer_norm<<<blocks, 256, 0, s>>>(dst, x, (int)ne0, nrows, m, eps, w, b, rms ? 1 : 0);
}
void launch_layer_norm_4d(hipStream_t s, float* dst, const float* x, const int64_t ne[4], const int64_t xnb[4], const int64_t dnb[4], float eps, const float* w,
                          const float* b, bool rms) {
    KScope ks_(s, KF_NORM_F32, 0.0, (double)ne[0] * ne[1] * ne[2] * ne[3] * 8.0);
    const int64_t nrows = ne[1] * ne[2] * ne[3];
    RowMap m{ne[1], ne[2], xnb[1] / 4, xnb[2] / 4, xnb[3] / 4, dnb[1] / 4, dnb[2] / 4, dnb[3] / 4};
    k_layer_norm<<<(int)((nrows + 3) / 4), 256, 0, s>>>(dst, x, (int)ne[0], nrows, m, eps, w, b, rms ? 1 : 0);
}

// one block per row; mask (f16 or f32) row selected as row % rows_per_mat (broadcast over batch)
__global__ __launch_bounds__(256) void k_soft_max(float* __restrict__ dst, const float* __restrict__ x, int ncols, float scale, const char* __restrict__ mask,
                                                  int mask_type, int64_t mask_nb1, int64_t rows_per_mat) {
    __shared__ float scratch[8];
    const int64_t row = blockIdx.x;
    const float* xr   = x + row * ncols;
    float* yr         = dst + row * ncols;
    const char* mr    = mask ? mask + (row % rows_per_mat) * mask_nb1 : nullptr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < ncols; i += 256) {
        float v = xr[i] * scale;
        if (mr) v += mask_type == 1 ? __half2float(((const __half*)mr)[i]) : ((const float*)mr)[i];
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    if (lane == 0) scratch[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < ncols; i += 256) {
        float v = xr[i] * scale;
        if (mr) v += mask_type == 1 ? __half2float(((const __half*)mr)[i]) : ((const float*)mr)[i];
        const float e = __expf(v - mx);
        yr[i]         = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) scratch[4 + wv] = sum;
    __syncthreads();
    const float inv = 1.0f / (scratch[4] + scratch[5] + scratch[6] + scratch[7]);
    for (int i = threadIdx.x; i < ncols; i += 256) yr[i] *= inv;
}

// row softmax of f32 scores written as the f16 operand image of the following P.V GEMM ([rows][Kp], zero padded): one block per row,
// float4 reads (the row is re-read from L2 for the sum and the write), 8-byte stores
__global__ __launch_bounds__(256) void k_soft_max_rows_f16(_Float16* __restrict__ dst, const float* __restrict__ x, int ncols, int Kp) {
    __shared__ float scratch[8];
    const int64_t row = blockIdx.x;
    const float4* xr  = (const float4*)(x + row * ncols);
    _Float16* yr      = dst + row * Kp;
    const int n4 = ncols / 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = xr[i];
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    mx = wave_max(mx);
    if (lane == 0) scratch[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    float sum = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = xr[i];
        sum += (__expf(v.x - mx) + __expf(v.y - mx)) + (__expf(v.z - mx) + __expf(v.w - mx));
    }
    sum = wave_sum(sum);
    if (lane == 0) scratch[4 + wv] = sum;
    __syncthreads();
    const float inv = 1.0f / (scratch[4] + scratch[5] + scratch[6] + scratch[7]);
    for (int i = threadIdx.x; i < Kp / 4; i += 256) {
        half4_t h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (i < n4) {
            const float4 v = xr[i];
            h[0] = (_Float16)(__expf(v.x - mx) * inv);
            h[1] = (_Float16)(__expf(v.y - mx) * inv);
            h[2] = (_Float16)(__expf(v.z - mx) * inv);
            h[3] = (_Float16)(__expf(v.w - mx) * inv);
        }
        *(half4_t*)(yr + i * 4) = h;
    }
}
void launch_soft_max_rows_f16(hipStream_t s, void* dst16, const float* x, int64_t ncols, int64_t nrows) {
    KScope ks_(s, KF_SOFTMAX, 0.0, (double)ncols * nrows * 6.0);
    const int Kp = (int)((ncols + 63) / 64 * 64);
    k_soft_max_rows_f16<<<(unsigned)nrows, 256, 0, s>>>((_Float16*)dst16, x, (int)ncols, Kp);
}

void launch_soft_max(hipStream_t s, float* dst, const float* x, int64_t ncols, int64_t nrows, float scale, const View4* mask, int64_t rows_per_mat) {
    KScope ks_(s, KF_SOFTMAX, 0.0, (double)ncols * nrows * 8.0);
    k_soft_max<<<(unsigned)nrows, 256, 0, s>>>(dst, x, (int)ncols, scale, mask ? (const char*)mask->data : nullptr, mask ? mask->type : 0,
                                             mask ? mask->nb[1] : 0, rows_per_mat);
}

}  // namespace mi355x
