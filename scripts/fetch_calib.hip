// fetch_calib.hip — known-byte-count calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (VERDICT r4 task 5b; MI355X_MICROARCH.md "HBM":
// "Other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel streams N bytes (default 1 GiB: 4x the 256 MiB Infinity Cache) exactly once, coalesced, with one per-lane access width:
//   k_read<4|8|16>   global loads of 4 / 8 / 16 bytes per lane (the conv epilogue's f32 residual reads are the 4-byte case the round-4 traffic
//                    ratio hinged on), k_read_lds16 = global_load_lds of 16 bytes per lane (the GEMM operand stream)
//   k_write<4|8|16>  global stores of that width (the f32 NCHW output stores of the conv epilogue are 4 bytes per lane)
//   k_rmw4           f32 read + f32 write of the same address range (residual add in place)
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib scripts/fetch_calib.hip ; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
// `--pmc WRITE_SIZE` (separate passes); scripts/fetch_calib.py divides the counters by the known bytes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ p, size_t n, float* sink) {
    float acc      = 0.f;
    const size_t s = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) {
        T v;
        if constexpr (sizeof(T) == 4) v = __builtin_nontemporal_load(p + i);
        else v = p[i];
        const float* f = (const float*)&v;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc += f[k];
    }
    if (acc == 123456.789f) *sink = acc;  // never true: keeps the loads
}
template <typename T>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ p, size_t n, float val) {
    const size_t s = (size_t)gridDim.x * blockDim.x;
    T v;
    float* f = (float*)&v;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) f[k] = val + k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) p[i] = v;
}
__global__ __launch_bounds__(256) void k_rmw4(float* __restrict__ p, size_t n) {
    const size_t s = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) p[i] = p[i] + 1.0f;
}
// LDS-DMA: each wave moves 1 KiB per instruction (16 bytes per lane, lane-linear LDS destination)
__global__ __launch_bounds__(256) void k_read_lds16(const float4* __restrict__ p, size_t n, float* sink) {
    __shared__ float4 buf[256];
    const size_t s = (size_t)gridDim.x * blockDim.x;
    const int wave = threadIdx.x >> 6;
    float acc      = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i), (__attribute__((address_space(3))) void*)(buf + wave * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[threadIdx.x].x;
    }
    if (acc == 123456.789f) *sink = acc;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1024) << 20;  // MiB
    const int reps     = argc > 2 ? atoi(argv[2]) : 3;
    void* buf;
    float* sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, bytes));
    const int grid = 256 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, auto launch) {
        launch();  // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s %zu bytes per launch  %8.1f us  %7.1f GB/s\n", name, bytes, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e9);
    };
    timed("k_read4", [&] { k_read<float><<<grid, 256>>>((const float*)buf, bytes / 4, sink); });
    timed("k_read8", [&] { k_read<float2><<<grid, 256>>>((const float2*)buf, bytes / 8, sink); });
    timed("k_read16", [&] { k_read<float4><<<grid, 256>>>((const float4*)buf, bytes / 16, sink); });
    timed("k_read_lds16", [&] { k_read_lds16<<<grid, 256>>>((const float4*)buf, bytes / 16, sink); });
    timed("k_write4", [&] { k_write<float><<<grid, 256>>>((float*)buf, bytes / 4, 1.f); });
    timed("k_write8", [&] { k_write<float2><<<grid, 256>>>((float2*)buf, bytes / 8, 1.f); });
    timed("k_write16", [&] { k_write<float4><<<grid, 256>>>((float4*)buf, bytes / 16, 1.f); });
    timed("k_rmw4", [&] { k_rmw4<<<grid, 256>>>((float*)buf, bytes / 4); });
    CK(hipFree(buf));
    return 0;
}
