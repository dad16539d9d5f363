// device_utils.h — small device-side helpers shared by the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace mi355x {

struct bf16_t {
    uint16_t bits;
};

template <typename T, int N>
struct vec_sel;
template <>
struct vec_sel<float, 4> {
    using type = float4;
};
template <>
struct vec_sel<__half, 4> {
    using type = uint2;
};
template <>
struct vec_sel<bf16_t, 4> {
    using type = uint2;
};
template <typename T, int N>
using vec_t = typename vec_sel<T, N>::type;

template <typename TD, typename TS>
__device__ __forceinline__ TD cvt(TS v);
template <>
__device__ __forceinline__ float cvt<float, float>(float v) { return v; }
template <>
__device__ __forceinline__ __half cvt<__half, float>(float v) { return __float2half_rn(v); }
template <>
__device__ __forceinline__ float cvt<float, __half>(__half v) { return __half2float(v); }
template <>
__device__ __forceinline__ __half cvt<__half, __half>(__half v) { return v; }
template <>
__device__ __forceinline__ float cvt<float, bf16_t>(bf16_t v) { return __uint_as_float((uint32_t)v.bits << 16); }
template <>
__device__ __forceinline__ bf16_t cvt<bf16_t, float>(float f) {
    uint32_t u = __float_as_uint(f);
    bf16_t r;
    if ((u & 0x7FFFFFFFu) > 0x7F800000u)
        r.bits = (uint16_t)((u >> 16) | 64);
    else
        r.bits = (uint16_t)((u + (0x7FFFu + ((u >> 16) & 1))) >> 16);
    return r;
}
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__half v) { return __half2float(v); }
__device__ __forceinline__ float to_f32(bf16_t v) { return cvt<float, bf16_t>(v); }

// activations.  GELU is the tanh approximation evaluated in f32 (ggml-cpu goes through an f16 table,
// i.e. the reference result carries an extra f16 rounding of input and output — covered by the tolerance).
// Reciprocals are v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the GEGLU / SiLU epilogues are VALU-bound.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
template <int OP>
__device__ __forceinline__ float act_apply(float x) {
    if (OP == UN_SILU) return x * fast_rcp(1.0f + __expf(-x));
    if (OP == UN_GELU) {
        // 0.5 x (1 + tanh(u)) with tanh(u) = 1 - 2/(exp(2u)+1)  ==  x * (1 - 1/(exp(2u)+1));  exp -> inf gives x, exp -> 0 gives 0
        const float u = 0.79788456080286535588f * x * (1.0f + 0.044715f * x * x);
        return x * (1.0f - fast_rcp(__expf(2.0f * u) + 1.0f));
    }
    if (OP == UN_GELU_QUICK) return x * fast_rcp(1.0f + __expf(-1.702f * x));
    if (OP == UN_SIGMOID) return fast_rcp(1.0f + __expf(-x));
    if (OP == UN_TANH) return tanhf(x);
    if (OP == UN_RELU) return x > 0.f ? x : 0.f;
    if (OP == UN_NEG) return -x;
    return __expf(x);
}
__device__ __forceinline__ float act_dyn(int op, float x) {
    switch (op) {
        case UN_SILU: return act_apply<UN_SILU>(x);
        case UN_GELU: return act_apply<UN_GELU>(x);
        case UN_GELU_QUICK: return act_apply<UN_GELU_QUICK>(x);
        case UN_SIGMOID: return act_apply<UN_SIGMOID>(x);
        case UN_TANH: return act_apply<UN_TANH>(x);
        case UN_RELU: return act_apply<UN_RELU>(x);
        case UN_NEG: return -x;
        case UN_EXP: return __expf(x);
        default: return x;
    }
}

// wave64 reductions via DPP-free shuffles (compiler lowers __shfl_xor to ds_swizzle / dpp where it can)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block reduction of two values with NW waves; scratch must hold 2*NW floats
template <int NW>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* scratch) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        scratch[w]      = a;
        scratch[NW + w] = b;
    }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        sa += scratch[i];
        sb += scratch[NW + i];
    }
    a = sa;
    b = sb;
    __syncthreads();
}

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));

}  // namespace mi355x
