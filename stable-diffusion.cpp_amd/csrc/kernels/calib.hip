// calib.hip — what THIS box delivers, measured in about a second: bench.py puts it next to the vendor peaks so that numbers of different runs can be
// compared (the pool's boxes differ by +-7 % with one binary: VERDICT r5 weak #12; SURVEY.md section 8(d): "re-measure on the box with a stream-triad and an
// MFMA-loop microbench; report both").
//   mfma_f16_tflops   v_mfma_f32_32x32x16_f16 issued back to back from registers: 8 waves per CU (2 per SIMD), 4 independent accumulators per wave, no memory
//                     traffic — the ceiling of the matrix pipe at the clock the chip sustains under that load
//   mfma_clock_mhz    shader clock during that loop: s_memtime (core clock) against s_memrealtime (constant 100 MHz)
//   copy_tbs          float4 copy of 1 GiB (read + write bytes / time), 256 threads x 4 float4 per thread in flight
//   read_tbs          float4 read-only pass over the same buffer (sum folded into one store per workgroup)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "kernels.h"

namespace mi355x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k_calib_mfma(float* sink, unsigned long long* clocks, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (float)((threadIdx.x + i) & 7));
        b[i] = (_Float16)(0.002f * (float)((threadIdx.x * 3 + i) & 7));
    }
    float16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;  // keeps the accumulators alive; never true
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = r1 - r0;
    }
}

__global__ __launch_bounds__(256) void k_calib_copy(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n4; i += 4 * stride) {
        const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        dst[i] = v0, dst[i + stride] = v1, dst[i + 2 * stride] = v2, dst[i + 3 * stride] = v3;
    }
}
__global__ __launch_bounds__(256) void k_calib_read(float* __restrict__ out, const float4* __restrict__ src, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n4; i += 4 * stride) {
        const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        acc += v0.x + v1.y + v2.z + v3.w;
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

}  // namespace

bool calibrate_device(hipStream_t s, CalibrationResult* out) {
    *out = CalibrationResult{};
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    float* sink              = nullptr;
    unsigned long long* clk  = nullptr;
    const size_t bytes       = (size_t)1 << 30;
    float4 *a = nullptr, *b = nullptr;
    bool ok = hipMalloc(&sink, 4096 * sizeof(float)) == hipSuccess && hipMalloc(&clk, 16) == hipSuccess && hipMalloc(&a, bytes) == hipSuccess && hipMalloc(&b, bytes) == hipSuccess;
    if (ok) {
        (void)hipMemsetAsync(a, 0, bytes, s);
        (void)hipMemsetAsync(b, 0, bytes, s);
        // ---- matrix pipe: one workgroup of 8 waves per CU
        const int iters = 20000;
        k_calib_mfma<<<cus, 512, 0, s>>>(sink, clk, 2000);  // warm-up: clocks ramp
        float best_ms = 1e30f;
        unsigned long long hc[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, s);
            k_calib_mfma<<<cus, 512, 0, s>>>(sink, clk, iters);
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best_ms) {
                best_ms = ms;
                (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
            }
        }
        const double flops   = (double)cus * 8.0 * iters * 16.0 * (2.0 * 32 * 32 * 16);
        out->mfma_f16_tflops = (float)(flops / (best_ms * 1e-3) / 1e12);
        out->mfma_clock_mhz  = hc[1] ? (float)((double)hc[0] / (double)hc[1] * 100.0) : 0.f;
        // ---- HBM: copy and read of 1 GiB
        const size_t n4 = bytes / 16;
        const int grid  = cus * 16;
        for (int pass = 0; pass < 2; ++pass) {
            float bm = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, s);
                if (pass == 0)
                    k_calib_copy<<<grid, 256, 0, s>>>(b, a, n4);
                else
                    k_calib_read<<<grid, 256, 0, s>>>(sink, a, n4);
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < bm) bm = ms;
            }
            const double moved = (pass == 0 ? 2.0 : 1.0) * (double)bytes;
            (pass == 0 ? out->copy_tbs : out->read_tbs) = (float)(moved / (bm * 1e-3) / 1e12);
        }
        out->compute_units = cus;
        ok                 = hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess;
    }
    if (sink) (void)hipFree(sink);
    if (clk) (void)hipFree(clk);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ok;
}

}  // namespace mi355x
