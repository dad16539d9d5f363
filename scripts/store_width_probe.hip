// store_width_probe.hip — what one global store / load instruction costs by per-lane width (round 4).
// Question: the short-K Linear epilogues (GEGLU / head-major f16: 2 bytes per lane per store, f32 + residual: 4 bytes per lane) look bound by the
// NUMBER of vector-memory instructions, not by bytes.  Every variant below moves the same byte count with full-line-coalesced accesses; only the
// per-lane width differs.  Also: the pattern of the transposed-accumulator epilogue (gemm16_swp: 16 bytes per lane, 32 rows x 32 bytes per
// instruction) and the MFMA D[row][col] f16 pattern (2 rows x 64 bytes per instruction, row stride ldd).
// build: hipcc -O3 --offload-arch=gfx950 scripts/store_width_probe.hip -o /tmp/store_width_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <typename T> __device__ T mk(float v);
template <> __device__ _Float16 mk<_Float16>(float v) { return (_Float16)v; }
template <> __device__ half2_t mk<half2_t>(float v) { return half2_t{(_Float16)v, (_Float16)v}; }
template <> __device__ float mk<float>(float v) { return v; }
template <> __device__ half4_t mk<half4_t>(float v) { return half4_t{(_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v}; }
template <> __device__ float2 mk<float2>(float v) { return make_float2(v, v); }
template <> __device__ float4 mk<float4>(float v) { return make_float4(v, v, v, v); }

// each workgroup owns a contiguous slab of `per_wg` bytes and writes it with NI = per_wg / (512 * sizeof(T)) fully coalesced instructions per thread
template <typename T>
__global__ __launch_bounds__(512) void k_store(char* out, size_t per_wg, float v) {
    T* p          = (T*)(out + (size_t)blockIdx.x * per_wg) + threadIdx.x;
    const int ni  = (int)(per_wg / (512 * sizeof(T)));
    const T val   = mk<T>(v + threadIdx.x);
#pragma unroll 8
    for (int i = 0; i < ni; ++i) p[(size_t)i * 512] = val;
}
template <typename T>
__global__ __launch_bounds__(512) void k_load(const char* in, size_t per_wg, float* sink) {
    const T* p   = (const T*)(in + (size_t)blockIdx.x * per_wg) + threadIdx.x;
    const int ni = (int)(per_wg / (512 * sizeof(T)));
    float acc    = 0.f;
#pragma unroll 8
    for (int i = 0; i < ni; ++i) {
        T x = p[(size_t)i * 512];
        acc += ((const _Float16*)&x)[0];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// MFMA D[row][col] f16 epilogue pattern: a wave stores 2 rows x 32 halfs (64 B) per instruction, 16 instructions cover a 32 x 32 block; row stride ld halfs
__global__ __launch_bounds__(512) void k_store_mfma16(_Float16* out, int ld, int nblk_per_wave, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, lc = lane & 31;
    // workgroup tile: 256 rows x (nblk_per_wave * 32) cols; wave w owns rows 32 w .. 32 w + 31
    _Float16* base = out + ((size_t)blockIdx.x * 256 + wave * 32) * ld;
    for (int cb = 0; cb < nblk_per_wave; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2) + 4 * hi;
            base[(size_t)ro * ld + cb * 32 + lc] = (_Float16)(v + r);
        }
}
// the same outputs written as 16-byte chunks in row order: thread t of the workgroup -> row t / (cols / 8), chunk t % (cols / 8) (what an LDS-staged epilogue would issue)
__global__ __launch_bounds__(512) void k_store_rows16(_Float16* out, int ld, int cols, float v) {
    const int cpr = cols / 8;  // chunks per row
    _Float16* base = out + (size_t)blockIdx.x * 256 * ld;
    const half8_t val = {(_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v};
    for (int c = threadIdx.x; c < 256 * cpr; c += 512) {
        const int row = c / cpr, j = c - row * cpr;
        *(half8_t*)(base + (size_t)row * ld + j * 8) = val;
    }
}
// transposed accumulator (gemm16_swp) f16 pattern: lane = row, 8 bytes (4 halfs) per lane: an instruction touches 32 rows x 16 bytes
__global__ __launch_bounds__(512) void k_store_swp16(_Float16* out, int ld, int nblk_per_wave, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, lr = lane & 31;
    _Float16* base = out + ((size_t)blockIdx.x * 256 + wave * 32 + lr) * ld;
    const half4_t val = {(_Float16)v, (_Float16)v, (_Float16)v, (_Float16)v};
    for (int cb = 0; cb < nblk_per_wave; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) *(half4_t*)(base + cb * 32 + 8 * q + 4 * hi) = val;
}

template <typename F>
static float time_ms(F f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t total = (size_t)512 << 20;  // 512 MiB: beyond the Infinity Cache
    char* buf;
    float* sink;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, total));
    for (int wgs : {1024, 4096}) {
        const size_t per = total / wgs;
        printf("== contiguous slabs, %d workgroups x 512 threads, %zu KiB each, 512 MiB per launch\n", wgs, per >> 10);
#define RUN_ST(T, name) { float ms = time_ms([&] { k_store<T><<<wgs, 512>>>(buf, per, 1.f); }); printf("  store %-10s %2zu B/lane: %7.1f us  %6.2f TB/s  %5.2f B/clk/CU @2.4GHz\n", name, sizeof(T), ms * 1e3, total / ms / 1e9, total / (ms * 1e-3) / 256 / 2.4e9); }
        RUN_ST(_Float16, "half");
        RUN_ST(half2_t, "half2");
        RUN_ST(float, "float");
        RUN_ST(float2, "float2");
        RUN_ST(float4, "float4");
#define RUN_LD(T, name) { float ms = time_ms([&] { k_load<T><<<wgs, 512>>>(buf, per, sink); }); printf("  load  %-10s %2zu B/lane: %7.1f us  %6.2f TB/s  %5.2f B/clk/CU @2.4GHz\n", name, sizeof(T), ms * 1e3, total / ms / 1e9, total / (ms * 1e-3) / 256 / 2.4e9); }
        RUN_LD(_Float16, "half");
        RUN_LD(float, "float");
        RUN_LD(float2, "float2");
        RUN_LD(float4, "float4");
    }
    // epilogue patterns: FF1 of the 64x64 SD1.5 level: 65536 rows x 1280 f16 outputs (ld 1280), 256-row workgroup tiles x 64 output columns
    {
        const int rows = 65536, ld = 1280;
        const size_t bytes = (size_t)rows * ld * 2;
        for (int cols : {64, 128}) {
            const int nblk = cols / 32;
            // simpler and closer to the kernel: the output is a [rows * ctiles][cols] matrix with ld = cols (contiguous tiles) for variant A, and the real strided layout for variant B
            for (int strided = 0; strided < 2; ++strided) {
                const int ctiles = ld / cols;
                const int l      = strided ? ld : cols;
                float t0 = time_ms([&] {
                    if (!strided) k_store_mfma16<<<rows / 256 * ctiles, 512>>>((_Float16*)buf, l, nblk, 1.f);
                    else for (int ct = 0; ct < ctiles; ++ct) k_store_mfma16<<<rows / 256, 512>>>((_Float16*)buf + ct * cols, l, nblk, 1.f);
                });
                float t1 = time_ms([&] {
                    if (!strided) k_store_rows16<<<rows / 256 * ctiles, 512>>>((_Float16*)buf, l, cols, 1.f);
                    else for (int ct = 0; ct < ctiles; ++ct) k_store_rows16<<<rows / 256, 512>>>((_Float16*)buf + ct * cols, l, cols, 1.f);
                });
                float t2 = time_ms([&] {
                    if (!strided) k_store_swp16<<<rows / 256 * ctiles, 512>>>((_Float16*)buf, l, nblk, 1.f);
                    else for (int ct = 0; ct < ctiles; ++ct) k_store_swp16<<<rows / 256, 512>>>((_Float16*)buf + ct * cols, l, nblk, 1.f);
                });
                printf("== f16 epilogue patterns, 65536 x 1280 outputs (%.0f MB), tile 256 x %d, %s: D[row][col] 2 B/lane %7.1f us (%.2f TB/s) | rows as 16-B chunks %7.1f us (%.2f TB/s) | swp 8 B/lane %7.1f us (%.2f TB/s)\n",
                       bytes / 1e6, cols, strided ? "row stride 1280 (one launch per column tile)" : "contiguous tiles", t0 * 1e3, bytes / t0 / 1e9, t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9);
            }
        }
    }
    return 0;
}
