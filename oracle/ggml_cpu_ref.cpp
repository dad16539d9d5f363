// =================================================================================================
// oracle/ggml_cpu_ref.cpp — TEST INFRASTRUCTURE ONLY.  *** PARITY UNPINNED ***
//
// CPU restatement of the arithmetic behind stable-diffusion.cpp's hot path: the ggml-cpu backend's
// implementation of every ggml op the UNet / MMDiT denoise graphs and the KL-VAE decode graph emit
// (SURVEY.md §2.3).  It is packaged exactly like the thing it restates — a ggml backend plug-in
// (`ggml_backend_init`, device "CPU-oracle") — so tests run the SAME cgraph through this library and
// through libggml-mi355x.so and compare results.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
// product never links or dlopens it: the MI355X engine fails loudly when its HIP backend is missing.
//
// Provenance / pinning status
//   * The arithmetic of the reference lives in the third-party module `ggml`
//     (submodule https://github.com/leejet/ggml.git at path ggml/, /root/reference/.gitmodules:1-3),
//     which is EMPTY in /root/reference and whose pinned commit is unrecoverable.
//     It cannot be compiled into oracle/_ref.  The reference has no tests, golden vectors or
//     fixtures for this path (CMakeLists.txt:82).  => parity unpinned.
//   * What this file follows instead: upstream ggml-cpu's published algorithms and rounding points
//     (SURVEY.md Appendix A / E), anchored on the reference's own call sites:
//       op constructors + wrappers     src/core/ggml_extend.hpp:953-1652
//       graph dispatch                 src/core/ggml_extend_backend.cpp:466-509
//       load-time quantisation         src/model_loader.cpp:160-205
//   * Independent pin that IS available here: tests/golden/*.npz are produced by PyTorch-CPU fp32
//     (tests/golden/make_golden.py) for every op below and this oracle is checked against them
//     (tests/test_oracle_ops.py); whole-model graphs are checked against oracle/torch_ref.py.
//
// Rounding points reproduced (Appendix E):
//   MUL_MAT  F16 src0: src1 row -> F16, f32 accumulate.   BF16: src1 -> BF16.
//            Q8_0/Q4_0 src0: src1 row -> Q8_0 blocks (d stored as f16), int dot, * d0*d1, f32 accumulate.
//   IM2COL   emits F16 (conv = F16 x F16 -> f32).
//   FLASH_ATTN_EXT  q row -> F16, K.q f16 dot, online softmax f32, V accumulation in F16 when V is F16.
//   GELU / GELU_QUICK through a 65536-entry F16 table; SILU / SIGMOID in f32.
//   NORM / RMS_NORM / GROUP_NORM / SOFT_MAX sums in double.
// =================================================================================================
#include <immintrin.h>
#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ggml-abi.h"

namespace {

// ---------------------------------------------------------------- fp16 / bf16
inline float h2f(ggml_fp16_t h) { return _cvtsh_ss(h); }
inline ggml_fp16_t f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT); }
inline float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 64);
    return (uint16_t)((u + (0x7FFFu + ((u >> 16) & 1))) >> 16);
}

ggml_fp16_t g_gelu_table[65536];
ggml_fp16_t g_gelu_quick_table[65536];
const float GELU_COEF_A     = 0.044715f;
const float SQRT_2_OVER_PI  = 0.79788456080286535587989211986876f;
const float GELU_QUICK_COEF = -1.702f;
inline float gelu_f32(float x) { return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x))); }
inline float gelu_quick_f32(float x) { return x * (1.0f / (1.0f + expf(GELU_QUICK_COEF * x))); }
// Threads the oracle uses: the CPUs this process may actually run on (affinity mask and cgroup CPU quota), capped at 16
// — GPU boxes report hundreds of hardware threads to a container that may only schedule a few of them, and an
// oversubscribed OpenMP team spin-waiting at every one of the ~3k per-graph parallel regions is pathologically slow.
int usable_cpus(bool cap16 = true) {
    int n = omp_get_num_procs();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = atol(quota);
            if (q > 0) n = std::min<int>(n, (int)std::max<long>(1, (q + period - 1) / period));
        }
        fclose(f);
    }
    if (!cap16) return std::max(1, n);  // every CPU this process may be scheduled on (oracle_set_num_threads(0): bench.py's all-core cpu_baseline figure)
    if (const char* e = getenv("ORACLE_THREADS")) n = std::max(1, atoi(e));
    else {
        n = std::min(n, 16);
        // an explicit OMP_NUM_THREADS is a ceiling (several oracle processes sharing a box: tests/test_dist_shard.py)
        if (const char* o = getenv("OMP_NUM_THREADS"))
            if (atoi(o) > 0) n = std::min(n, atoi(o));
    }
    return std::max(1, n);
}

void init_tables() {
    static bool done = false;
    if (done) return;
    omp_set_dynamic(0);
    omp_set_num_threads(usable_cpus());
    for (int i = 0; i < 65536; ++i) {
        const float f         = h2f((ggml_fp16_t)i);
        g_gelu_table[i]       = f2h(gelu_f32(f));
        g_gelu_quick_table[i] = f2h(gelu_quick_f32(f));
    }
    done = true;
}

// ---------------------------------------------------------------- helpers
inline const char* cptr(const ggml_tensor* t) { return (const char*)t->data; }
inline char* mptr(ggml_tensor* t) { return (char*)t->data; }
inline float opf(const ggml_tensor* t, int i) { return ggml_abi_op_param_f32(t, i); }

float load_as_f32(const ggml_tensor* t, const char* p) {
    switch (t->type) {
        case GGML_TYPE_F32: return *(const float*)p;
        case GGML_TYPE_F16: return h2f(*(const ggml_fp16_t*)p);
        case GGML_TYPE_BF16: return bf2f(*(const uint16_t*)p);
        case GGML_TYPE_I32: return (float)*(const int32_t*)p;
        default: return 0.0f;
    }
}

// ---------------------------------------------------------------- Q8_0 activation quantisation
struct q8blk {
    float d;  // already rounded through f16, as ggml stores it
    int8_t qs[32];
};
void quantize_row_q8_0(const float* x, q8blk* y, int64_t k) {
    for (int64_t i = 0; i < k / 32; ++i) {
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) amax = std::max(amax, fabsf(x[i * 32 + j]));
        const float d  = amax / 127.0f;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d         = h2f(f2h(d));
        for (int j = 0; j < 32; ++j) y[i].qs[j] = (int8_t)roundf(x[i * 32 + j] * id);
    }
}

// ---------------------------------------------------------------- f32 dot-product micro kernels
// C[m][n] = sum_k A[m][k]*B[n][k], both K-contiguous f32; f32 accumulation in 8-lane partial sums
// (the shape of ggml-cpu's SIMD vec_dot); 2x4 register block.
inline float hsum8(__m256 v) {
    __m128 lo = _mm256_castps256_ps128(v), hi = _mm256_extractf128_ps(v, 1);
    lo        = _mm_add_ps(lo, hi);
    lo        = _mm_hadd_ps(lo, lo);
    lo        = _mm_hadd_ps(lo, lo);
    return _mm_cvtss_f32(lo);
}
// one 2 x 4 block of outputs: 8-lane FMA chains over k, horizontal sum, scalar tail — the per-element arithmetic (and therefore every output bit) is independent
// of how the blocks are scheduled below
static inline void gemm_nt_2x4(const float* a0, const float* a1, const float* b0, const float* b1, const float* b2, const float* b3, int64_t K, float (&out)[2][4]) {
    __m256 acc[2][4];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b) acc[a][b] = _mm256_setzero_ps();
    int64_t k = 0;
    for (; k + 8 <= K; k += 8) {
        const __m256 va0 = _mm256_loadu_ps(a0 + k), va1 = _mm256_loadu_ps(a1 + k);
        const __m256 vb0 = _mm256_loadu_ps(b0 + k), vb1 = _mm256_loadu_ps(b1 + k);
        const __m256 vb2 = _mm256_loadu_ps(b2 + k), vb3 = _mm256_loadu_ps(b3 + k);
        acc[0][0] = _mm256_fmadd_ps(va0, vb0, acc[0][0]);
        acc[0][1] = _mm256_fmadd_ps(va0, vb1, acc[0][1]);
        acc[0][2] = _mm256_fmadd_ps(va0, vb2, acc[0][2]);
        acc[0][3] = _mm256_fmadd_ps(va0, vb3, acc[0][3]);
        acc[1][0] = _mm256_fmadd_ps(va1, vb0, acc[1][0]);
        acc[1][1] = _mm256_fmadd_ps(va1, vb1, acc[1][1]);
        acc[1][2] = _mm256_fmadd_ps(va1, vb2, acc[1][2]);
        acc[1][3] = _mm256_fmadd_ps(va1, vb3, acc[1][3]);
    }
    float tail[2][4] = {{0}};
    for (; k < K; ++k) {
        const float bv[4] = {b0[k], b1[k], b2[k], b3[k]};
        for (int b = 0; b < 4; ++b) {  // explicit fused multiply-adds: the scalar tail must not depend on what the compiler contracts (K % 8 != 0: conv_in's K = 36)
            tail[0][b] = fmaf(a0[k], bv[b], tail[0][b]);
            tail[1][b] = fmaf(a1[k], bv[b], tail[1][b]);
        }
    }
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b) out[a][b] = hsum8(acc[a][b]) + tail[a][b];
}
// AVX-512 hosts (the GPU boxes' EPYC 9575F, this container's Xeon): a 4 x 4 block on 16-lane FMA chains — what ggml-cpu's vec_dot does there
// (GGML_F32_EPR = 16).  Per-element arithmetic: 16 strided partial sums over k (the last K % 16 elements in one masked step), a tree reduction.  It differs from the AVX2
// block above by summation order only (~1e-7 relative); chosen at run time, ORACLE_NO_AVX512=1 forces the AVX2 block.  ~2-3x the AVX2 block's rate: the
// whole-model oracle forwards of the full-width / full-depth GPU tests are what the GPU suite's wall time consists of.
__attribute__((target("avx512f"))) static inline void gemm_nt_4x4_avx512(const float* const (&a)[4], const float* const (&b)[4], int64_t K, float (&out)[4][4]) {
    __m512 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = _mm512_setzero_ps();
    int64_t k = 0;
    for (; k + 16 <= K; k += 16) {
        const __m512 va0 = _mm512_loadu_ps(a[0] + k), va1 = _mm512_loadu_ps(a[1] + k), va2 = _mm512_loadu_ps(a[2] + k), va3 = _mm512_loadu_ps(a[3] + k);
        for (int j = 0; j < 4; ++j) {
            const __m512 vb = _mm512_loadu_ps(b[j] + k);
            acc[0][j]       = _mm512_fmadd_ps(va0, vb, acc[0][j]);
            acc[1][j]       = _mm512_fmadd_ps(va1, vb, acc[1][j]);
            acc[2][j]       = _mm512_fmadd_ps(va2, vb, acc[2][j]);
            acc[3][j]       = _mm512_fmadd_ps(va3, vb, acc[3][j]);
        }
    }
    if (k < K) {  // the last K % 16 elements: one masked step on the same lane chains (attention's K = d_head = 40 would otherwise run 128 scalar fmas per block)
        const __mmask16 m = (__mmask16)((1u << (K - k)) - 1u);
        const __m512 va0 = _mm512_maskz_loadu_ps(m, a[0] + k), va1 = _mm512_maskz_loadu_ps(m, a[1] + k), va2 = _mm512_maskz_loadu_ps(m, a[2] + k), va3 = _mm512_maskz_loadu_ps(m, a[3] + k);
        for (int j = 0; j < 4; ++j) {
            const __m512 vb = _mm512_maskz_loadu_ps(m, b[j] + k);
            acc[0][j]       = _mm512_fmadd_ps(va0, vb, acc[0][j]);
            acc[1][j]       = _mm512_fmadd_ps(va1, vb, acc[1][j]);
            acc[2][j]       = _mm512_fmadd_ps(va2, vb, acc[2][j]);
            acc[3][j]       = _mm512_fmadd_ps(va3, vb, acc[3][j]);
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[i][j] = _mm512_reduce_add_ps(acc[i][j]);
}
__attribute__((target("avx512f"))) static void gemm_nt_f32_avx512(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc_n, int64_t M, int64_t N, int64_t K) {
    const int64_t groups = (N + 3) / 4;
    const int nth        = std::max(1, omp_get_max_threads());
    int64_t per          = groups / ((int64_t)nth * 4);
    per                  = std::max<int64_t>(1, std::min<int64_t>(per, K <= 4096 ? 12 : (K <= 8192 ? 6 : 3)));
    const int64_t NB = per * 4, nchunks = (N + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int64_t nbeg = ch * NB, nend = std::min(N, nbeg + NB);
        for (int64_t m0 = 0; m0 < M; m0 += 4) {
            const int64_t mm   = std::min<int64_t>(4, M - m0);
            const float* const a[4] = {A + m0 * lda, A + (m0 + (mm > 1 ? 1 : 0)) * lda, A + (m0 + (mm > 2 ? 2 : 0)) * lda, A + (m0 + (mm > 3 ? 3 : 0)) * lda};
            for (int64_t n0 = nbeg; n0 < nend; n0 += 4) {
                const int64_t nn   = std::min<int64_t>(4, N - n0);
                const float* const b[4] = {B + n0 * ldb, B + (n0 + (nn > 1 ? 1 : 0)) * ldb, B + (n0 + (nn > 2 ? 2 : 0)) * ldb, B + (n0 + (nn > 3 ? 3 : 0)) * ldb};
                float out[4][4];
                gemm_nt_4x4_avx512(a, b, K, out);
                for (int i = 0; i < mm; ++i)
                    for (int j = 0; j < nn; ++j) C[(n0 + j) * ldc_n + m0 + i] = out[i][j];
            }
        }
    }
}
static bool oracle_use_avx512() {
    static const bool on = __builtin_cpu_supports("avx512f") && !getenv("ORACLE_NO_AVX512");
    return on;
}
void gemm_nt_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc_n, int64_t M, int64_t N, int64_t K) {
    // C element (m, n) stored at C[n*ldc_n + m]  (ggml dst layout: ne0 = M contiguous)
    if (oracle_use_avx512()) {
        gemm_nt_f32_avx512(A, lda, B, ldb, C, ldc_n, M, N, K);
        return;
    }
    // Schedule (round 5; results bit-identical to the round-1 loop, which walked all of A once per FOUR rows of B and ran at DRAM speed): a thread takes a
    // chunk of NB rows of B (kept in its L2 / L3 slice), walks the rows of A in pairs ONCE per chunk and reuses each pair (L1-resident) for every group of
    // four B rows of the chunk.  ~3x faster on the whole-model oracle forwards the full-depth parity tests run.
    const int64_t groups = (N + 3) / 4;
    const int nth        = std::max(1, omp_get_max_threads());
    int64_t per          = groups / ((int64_t)nth * 4);  // >= 4 chunks per thread where N allows it (dynamic schedule evens out the ragged end)
    per                  = std::max<int64_t>(1, std::min<int64_t>(per, K <= 4096 ? 12 : (K <= 8192 ? 6 : 3)));  // chunk = 4 * per rows of B: <= ~600 KB
    const int64_t NB      = per * 4;
    const int64_t nchunks = (N + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int64_t nbeg = ch * NB, nend = std::min(N, nbeg + NB);
        for (int64_t m0 = 0; m0 < M; m0 += 2) {
            const int64_t mm = std::min<int64_t>(2, M - m0);
            const float* a0 = A + m0 * lda;
            const float* a1 = A + (m0 + (mm > 1 ? 1 : 0)) * lda;
            for (int64_t n0 = nbeg; n0 < nend; n0 += 4) {
                const int64_t nn = std::min<int64_t>(4, N - n0);
                const float* b0 = B + n0 * ldb;
                const float* b1 = B + (n0 + (nn > 1 ? 1 : 0)) * ldb;
                const float* b2 = B + (n0 + (nn > 2 ? 2 : 0)) * ldb;
                const float* b3 = B + (n0 + (nn > 3 ? 3 : 0)) * ldb;
                float out[2][4];
                gemm_nt_2x4(a0, a1, b0, b1, b2, b3, K, out);
                for (int a = 0; a < mm; ++a)
                    for (int b = 0; b < nn; ++b) C[(n0 + b) * ldc_n + m0 + a] = out[a][b];
            }
        }
    }
}

bool g_exact_weights = false;  // oracle_set_exact_weights()
// ---------------------------------------------------------------- MUL_MAT
// reference semantics: SURVEY.md Appendix A + E.1; call sites ggml_extend.hpp:1022,1028,1161,1467,1475
void op_mul_mat(ggml_tensor* dst) {
    const ggml_tensor* s0 = dst->src[0];
    const ggml_tensor* s1 = dst->src[1];
    const int64_t K = s0->ne[0], M = s0->ne[1], N = s1->ne[1];
    const int64_t ne02 = s0->ne[2], ne03 = s0->ne[3], ne12 = s1->ne[2], ne13 = s1->ne[3];
    const int64_t r2 = ne12 / ne02, r3 = ne13 / ne03;
    // g_exact_weights (oracle_set_exact_weights, tests only): the weight is widened to f32 EXACTLY (bf16 / f16 / dequantised q8_0 / q4_0 blocks) and the
    // activation row is NOT rounded to the weight's vec_dot type — the arithmetic-exact product both ggml-cpu's path and the GPU's approximate
    const bool exact = g_exact_weights && (s0->type == GGML_TYPE_Q8_0 || s0->type == GGML_TYPE_Q4_0 || s0->type == GGML_TYPE_BF16 || s0->type == GGML_TYPE_F16);
    const bool quant = !exact && (s0->type == GGML_TYPE_Q8_0 || s0->type == GGML_TYPE_Q4_0);

    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    std::vector<q8blk> Bq;
    if (quant) Bq.resize((size_t)N * (K / 32));

    for (int64_t i13 = 0; i13 < ne13; ++i13) {
        for (int64_t i12 = 0; i12 < ne12; ++i12) {
            const int64_t i03 = i13 / r3, i02 = i12 / r2;
            const char* a_base = cptr(s0) + i02 * s0->nb[2] + i03 * s0->nb[3];
            const char* b_base = cptr(s1) + i12 * s1->nb[2] + i13 * s1->nb[3];
            float* c_base      = (float*)(mptr(dst) + i12 * dst->nb[2] + i13 * dst->nb[3]);
            const int64_t ldc  = dst->nb[1] / sizeof(float);

            // --- src1 rows -> vec_dot_type of src0, then widened back to f32 for the FMA kernel
#pragma omp parallel for schedule(static)
            for (int64_t n = 0; n < N; ++n) {
                const char* row = b_base + n * s1->nb[1];
                float* out      = B.data() + n * K;
                std::vector<float> tmp;
                const float* xf = nullptr;
                if (s1->type == GGML_TYPE_F32 && s1->nb[0] == 4) {
                    xf = (const float*)row;
                } else {
                    tmp.resize(K);
                    for (int64_t k = 0; k < K; ++k) tmp[k] = load_as_f32(s1, row + k * s1->nb[0]);
                    xf = tmp.data();
                }
                switch (exact ? GGML_TYPE_F32 : s0->type) {
                    case GGML_TYPE_F16:
                        for (int64_t k = 0; k < K; ++k) out[k] = h2f(f2h(xf[k]));
                        break;
                    case GGML_TYPE_BF16:
                        for (int64_t k = 0; k < K; ++k) out[k] = bf2f(f2bf(xf[k]));
                        break;
                    case GGML_TYPE_Q8_0:
                    case GGML_TYPE_Q4_0:
                        quantize_row_q8_0(xf, Bq.data() + n * (K / 32), K);
                        break;
                    default:
                        for (int64_t k = 0; k < K; ++k) out[k] = xf[k];
                }
            }

            if (!quant) {
#pragma omp parallel for schedule(static)
                for (int64_t m = 0; m < M; ++m) {
                    const char* row = a_base + m * s0->nb[1];
                    float* out      = A.data() + m * K;
                    switch (s0->type) {
                        case GGML_TYPE_F16:
                            for (int64_t k = 0; k < K; ++k) out[k] = h2f(((const ggml_fp16_t*)row)[k]);
                            break;
                        case GGML_TYPE_BF16:
                            for (int64_t k = 0; k < K; ++k) out[k] = bf2f(((const uint16_t*)row)[k]);
                            break;
                        case GGML_TYPE_Q8_0:  // exact mode only: block = f16 d + 32 int8
                            for (int64_t b = 0; b < K / 32; ++b) {
                                ggml_fp16_t h;
                                memcpy(&h, row + b * 34, 2);
                                const float d = h2f(h);
                                for (int j = 0; j < 32; ++j) out[b * 32 + j] = d * (float)((const int8_t*)(row + b * 34 + 2))[j];
                            }
                            break;
                        case GGML_TYPE_Q4_0:  // exact mode only: block = f16 d + 16 bytes; element j < 16 = low nibble of qs[j], j >= 16 = high nibble of qs[j - 16]
                            for (int64_t b = 0; b < K / 32; ++b) {
                                ggml_fp16_t h;
                                memcpy(&h, row + b * 18, 2);
                                const float d     = h2f(h);
                                const uint8_t* qs = (const uint8_t*)(row + b * 18 + 2);
                                for (int j = 0; j < 16; ++j) {
                                    out[b * 32 + j]      = d * (float)((int)(qs[j] & 0xF) - 8);
                                    out[b * 32 + 16 + j] = d * (float)((int)(qs[j] >> 4) - 8);
                                }
                            }
                            break;
                        default:
                            for (int64_t k = 0; k < K; ++k) out[k] = ((const float*)row)[k];
                    }
                }
                gemm_nt_f32(A.data(), K, B.data(), K, c_base, ldc, M, N, K);
            } else {
                const int64_t nb   = K / 32;
                const size_t bsize = ggml_abi_type_size(s0->type);
#pragma omp parallel for schedule(static)
                for (int64_t m = 0; m < M; ++m) {
                    const uint8_t* row = (const uint8_t*)(a_base + m * s0->nb[1]);
                    // unpack the weight row once
                    std::vector<int8_t> wq(K);
                    std::vector<float> wd(nb);
                    for (int64_t b = 0; b < nb; ++b) {
                        ggml_fp16_t h;
                        memcpy(&h, row + b * bsize, 2);
                        wd[b] = h2f(h);
                        if (s0->type == GGML_TYPE_Q8_0) {
                            memcpy(wq.data() + b * 32, row + b * bsize + 2, 32);
                        } else {
                            const uint8_t* qs = row + b * bsize + 2;
                            for (int j = 0; j < 16; ++j) {
                                wq[b * 32 + j]      = (int8_t)((qs[j] & 0xF) - 8);
                                wq[b * 32 + 16 + j] = (int8_t)((qs[j] >> 4) - 8);
                            }
                        }
                    }
                    for (int64_t n = 0; n < N; ++n) {
                        const q8blk* y = Bq.data() + n * nb;
                        float sumf     = 0.0f;
                        for (int64_t b = 0; b < nb; ++b) {
                            int sumi = 0;
                            for (int j = 0; j < 32; ++j) sumi += (int)wq[b * 32 + j] * (int)y[b].qs[j];
                            sumf += (float)sumi * (wd[b] * y[b].d);
                        }
                        c_base[n * ldc + m] = sumf;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- IM2COL (F16 or F32 dst)
void op_im2col(ggml_tensor* dst) {
    const ggml_tensor* kern = dst->src[0];
    const ggml_tensor* x    = dst->src[1];
    const int32_t* p        = dst->op_params;
    const int s0 = p[0], s1 = p[1], p0 = p[2], p1 = p[3], d0 = p[4], d1 = p[5];
    const bool is_2D = p[6] == 1;
    const int64_t N = is_2D ? x->ne[3] : x->ne[2], IC = is_2D ? x->ne[2] : x->ne[1];
    const int64_t IH = is_2D ? x->ne[1] : 1, IW = x->ne[0];
    const int64_t KH = is_2D ? kern->ne[1] : 1, KW = kern->ne[0];
    const int64_t OH = is_2D ? dst->ne[2] : 1, OW = dst->ne[1];
    const size_t ofs0 = is_2D ? x->nb[3] : x->nb[2], ofs1 = is_2D ? x->nb[2] : x->nb[1];
    const int64_t CK = IC * KH * KW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t in = 0; in < N; ++in) {
        for (int64_t ioh = 0; ioh < OH; ++ioh) {
            for (int64_t iow = 0; iow < OW; ++iow) {
                char* drow = mptr(dst) + ((in * OH + ioh) * OW + iow) * dst->nb[1];
                for (int64_t iic = 0; iic < IC; ++iic) {
                    const char* src = cptr(x) + in * ofs0 + iic * ofs1;
                    for (int64_t ikh = 0; ikh < KH; ++ikh) {
                        for (int64_t ikw = 0; ikw < KW; ++ikw) {
                            const int64_t iiw = iow * s0 + ikw * d0 - p0;
                            const int64_t iih = ioh * s1 + ikh * d1 - p1;
                            float v           = 0.0f;
                            if (iih >= 0 && iih < IH && iiw >= 0 && iiw < IW) v = load_as_f32(x, src + iih * x->nb[1] + iiw * x->nb[0]);
                            const int64_t k = iic * KH * KW + ikh * KW + ikw;
                            if (dst->type == GGML_TYPE_F16)
                                ((ggml_fp16_t*)drow)[k] = f2h(v);
                            else
                                ((float*)drow)[k] = v;
                        }
                    }
                }
            }
        }
    }
    (void)CK;
}

// direct conv: same math as im2col(F16) x F16 kernel -> f32
void op_conv_2d(ggml_tensor* dst) {
    const ggml_tensor* kern = dst->src[0];
    const ggml_tensor* x    = dst->src[1];
    const int32_t* p        = dst->op_params;
    const int s0 = p[0], s1 = p[1], p0 = p[2], p1 = p[3], d0 = p[4], d1 = p[5];
    const int64_t KW = kern->ne[0], KH = kern->ne[1], IC = kern->ne[2], OC = kern->ne[3];
    const int64_t IW = x->ne[0], IH = x->ne[1], N = x->ne[3];
    const int64_t OW = dst->ne[0], OH = dst->ne[1];
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int64_t oc = 0; oc < OC; ++oc)
            for (int64_t oh = 0; oh < OH; ++oh)
                for (int64_t ow = 0; ow < OW; ++ow) {
                    float acc = 0.0f;
                    for (int64_t ic = 0; ic < IC; ++ic)
                        for (int64_t kh = 0; kh < KH; ++kh)
                            for (int64_t kw = 0; kw < KW; ++kw) {
                                const int64_t iw = ow * s0 + kw * d0 - p0, ih = oh * s1 + kh * d1 - p1;
                                if (iw < 0 || iw >= IW || ih < 0 || ih >= IH) continue;
                                const float xv = h2f(f2h(load_as_f32(x, cptr(x) + iw * x->nb[0] + ih * x->nb[1] + ic * x->nb[2] + n * x->nb[3])));
                                const float wv = load_as_f32(kern, cptr(kern) + kw * kern->nb[0] + kh * kern->nb[1] + ic * kern->nb[2] + oc * kern->nb[3]);
                                acc += xv * wv;
                            }
                    *(float*)(mptr(dst) + ow * dst->nb[0] + oh * dst->nb[1] + oc * dst->nb[2] + n * dst->nb[3]) = acc;
                }
}

// ---------------------------------------------------------------- norms
void op_norm(ggml_tensor* dst, bool rms) {
    const ggml_tensor* x = dst->src[0];
    const float eps      = opf(dst, 0);
    const int64_t ne0 = x->ne[0], nr = ggml_abi_nrows(x);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nr; ++r) {
        const int64_t i1 = r % x->ne[1], i2 = (r / x->ne[1]) % x->ne[2], i3 = r / (x->ne[1] * x->ne[2]);
        const float* xr = (const float*)(cptr(x) + i1 * x->nb[1] + i2 * x->nb[2] + i3 * x->nb[3]);
        float* yr       = (float*)(mptr(dst) + i1 * dst->nb[1] + i2 * dst->nb[2] + i3 * dst->nb[3]);
        if (rms) {
            double sum = 0.0;
            for (int64_t i = 0; i < ne0; ++i) sum += (double)(xr[i] * xr[i]);
            const float mean  = (float)(sum / ne0);
            const float scale = 1.0f / sqrtf(mean + eps);
            for (int64_t i = 0; i < ne0; ++i) yr[i] = xr[i] * scale;
        } else {
            double sum = 0.0;
            for (int64_t i = 0; i < ne0; ++i) sum += (double)xr[i];
            const float mean = (float)(sum / ne0);
            double sum2      = 0.0;
            for (int64_t i = 0; i < ne0; ++i) {
                const float v = xr[i] - mean;
                yr[i]         = v;
                sum2 += (double)(v * v);
            }
            const float variance = (float)(sum2 / ne0);
            const float scale    = 1.0f / sqrtf(variance + eps);
            for (int64_t i = 0; i < ne0; ++i) yr[i] *= scale;
        }
    }
}

// x.ne=[W,H,C,N]; groups of ceil(C/n_groups) channels; biased variance (ggml_extend.hpp:1502-1520)
void op_group_norm(ggml_tensor* dst) {
    const ggml_tensor* x = dst->src[0];
    const int n_groups   = dst->op_params[0];
    const float eps      = opf(dst, 1);
    const int64_t ne00 = x->ne[0], ne01 = x->ne[1], ne02 = x->ne[2], ne03 = x->ne[3];
    const int64_t cpg = (ne02 + n_groups - 1) / n_groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t i03 = 0; i03 < ne03; ++i03) {
        for (int64_t g = 0; g < n_groups; ++g) {
            const int64_t start = g * cpg;
            const int64_t end   = std::min<int64_t>(start + cpg, ne02);
            if (start >= end) continue;
            const int64_t step = end - start;
            double sum         = 0.0;
            for (int64_t i02 = start; i02 < end; ++i02)
                for (int64_t i01 = 0; i01 < ne01; ++i01) {
                    const float* xr = (const float*)(cptr(x) + i01 * x->nb[1] + i02 * x->nb[2] + i03 * x->nb[3]);
                    double sumr     = 0.0;
                    for (int64_t i = 0; i < ne00; ++i) sumr += (double)xr[i];
                    sum += sumr;
                }
            const float mean = (float)(sum / (ne00 * ne01 * step));
            double sum2      = 0.0;
            for (int64_t i02 = start; i02 < end; ++i02)
                for (int64_t i01 = 0; i01 < ne01; ++i01) {
                    const float* xr = (const float*)(cptr(x) + i01 * x->nb[1] + i02 * x->nb[2] + i03 * x->nb[3]);
                    float* yr       = (float*)(mptr(dst) + i01 * dst->nb[1] + i02 * dst->nb[2] + i03 * dst->nb[3]);
                    double sumr     = 0.0;
                    for (int64_t i = 0; i < ne00; ++i) {
                        const float v = xr[i] - mean;
                        yr[i]         = v;
                        sumr += (double)(v * v);
                    }
                    sum2 += sumr;
                }
            const float variance = (float)(sum2 / (ne00 * ne01 * step));
            const float scale    = 1.0f / sqrtf(variance + eps);
            for (int64_t i02 = start; i02 < end; ++i02)
                for (int64_t i01 = 0; i01 < ne01; ++i01) {
                    float* yr = (float*)(mptr(dst) + i01 * dst->nb[1] + i02 * dst->nb[2] + i03 * dst->nb[3]);
                    for (int64_t i = 0; i < ne00; ++i) yr[i] *= scale;
                }
        }
    }
}

// ---------------------------------------------------------------- elementwise
void op_unary(ggml_tensor* dst) {
    const ggml_tensor* x = dst->src[0];
    const int64_t n      = ggml_abi_nelements(x);
    const ggml_unary_op u = ggml_abi_get_unary_op(dst);
    // all unary ops on the path run on contiguous f32 (ggml_ext_gelu inserts a cont otherwise)
    const float* xs = (const float*)cptr(x);
    float* ys       = (float*)mptr(dst);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float v = xs[i];
        float r;
        switch (u) {
            case GGML_UNARY_OP_SILU: r = v / (1.0f + expf(-v)); break;
            case GGML_UNARY_OP_SIGMOID: r = 1.0f / (1.0f + expf(-v)); break;
            case GGML_UNARY_OP_TANH: r = tanhf(v); break;
            case GGML_UNARY_OP_RELU: r = v > 0.f ? v : 0.f; break;
            case GGML_UNARY_OP_GELU:
                if (v <= -10.0f) r = 0.0f;
                else if (v >= 10.0f) r = v;
                else r = h2f(g_gelu_table[f2h(v)]);
                break;
            case GGML_UNARY_OP_GELU_QUICK: r = h2f(g_gelu_quick_table[f2h(v)]); break;
            case GGML_UNARY_OP_NEG: r = -v; break;
            case GGML_UNARY_OP_EXP: r = expf(v); break;
            default: r = NAN;
        }
        ys[i] = r;
    }
}

void op_binary(ggml_tensor* dst) {
    const ggml_tensor* a = dst->src[0];
    const ggml_tensor* b = dst->src[1];
    const int64_t nr     = ggml_abi_nrows(a);
    const enum ggml_op op = dst->op;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nr; ++r) {
        const int64_t i1 = r % a->ne[1], i2 = (r / a->ne[1]) % a->ne[2], i3 = r / (a->ne[1] * a->ne[2]);
        const int64_t j1 = i1 % b->ne[1], j2 = i2 % b->ne[2], j3 = i3 % b->ne[3];
        const char* ar = cptr(a) + i1 * a->nb[1] + i2 * a->nb[2] + i3 * a->nb[3];
        const char* br = cptr(b) + j1 * b->nb[1] + j2 * b->nb[2] + j3 * b->nb[3];
        char* dr       = mptr(dst) + i1 * dst->nb[1] + i2 * dst->nb[2] + i3 * dst->nb[3];
        for (int64_t i = 0; i < a->ne[0]; ++i) {
            const float x = load_as_f32(a, ar + i * a->nb[0]);
            const float y = load_as_f32(b, br + (i % b->ne[0]) * b->nb[0]);
            float v;
            switch (op) {
                case GGML_OP_ADD: v = x + y; break;
                case GGML_OP_SUB: v = x - y; break;
                case GGML_OP_MUL: v = x * y; break;
                default: v = x / y;
            }
            *(float*)(dr + i * dst->nb[0]) = v;
        }
    }
}

void op_scale(ggml_tensor* dst) {
    const ggml_tensor* x = dst->src[0];
    const float s = opf(dst, 0), b = opf(dst, 1);
    const int64_t nr = ggml_abi_nrows(x);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nr; ++r) {
        const int64_t i1 = r % x->ne[1], i2 = (r / x->ne[1]) % x->ne[2], i3 = r / (x->ne[1] * x->ne[2]);
        const float* xr = (const float*)(cptr(x) + i1 * x->nb[1] + i2 * x->nb[2] + i3 * x->nb[3]);
        float* yr       = (float*)(mptr(dst) + i1 * dst->nb[1] + i2 * dst->nb[2] + i3 * dst->nb[3]);
        for (int64_t i = 0; i < x->ne[0]; ++i) yr[i] = xr[i] * s + b;
    }
}

// DUP / CONT / CPY: logical-order copy with f32<->f16<->bf16 conversion
void store_from_f32(ggml_tensor* t, char* p, float v) {
    switch (t->type) {
        case GGML_TYPE_F32: *(float*)p = v; break;
        case GGML_TYPE_F16: *(ggml_fp16_t*)p = f2h(v); break;
        case GGML_TYPE_BF16: *(uint16_t*)p = f2bf(v); break;
        case GGML_TYPE_I32: *(int32_t*)p = (int32_t)v; break;
        default: break;
    }
}
void op_copy(ggml_tensor* dst) {
    const ggml_tensor* s = dst->src[0];
    const int64_t n      = ggml_abi_nelements(s);
    const int64_t s0 = s->ne[0], s1 = s->ne[1], s2 = s->ne[2];
    const int64_t d0 = dst->ne[0], d1 = dst->ne[1], d2 = dst->ne[2];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t a0 = i % s0, a1 = (i / s0) % s1, a2 = (i / (s0 * s1)) % s2, a3 = i / (s0 * s1 * s2);
        const int64_t b0 = i % d0, b1 = (i / d0) % d1, b2 = (i / (d0 * d1)) % d2, b3 = i / (d0 * d1 * d2);
        const char* sp = cptr(s) + a0 * s->nb[0] + a1 * s->nb[1] + a2 * s->nb[2] + a3 * s->nb[3];
        char* dp       = mptr(dst) + b0 * dst->nb[0] + b1 * dst->nb[1] + b2 * dst->nb[2] + b3 * dst->nb[3];
        if (s->type == dst->type) {
            memcpy(dp, sp, ggml_abi_type_size(s->type));
        } else {
            store_from_f32(dst, dp, load_as_f32(s, sp));
        }
    }
}

void op_concat(ggml_tensor* dst) {
    const ggml_tensor* a = dst->src[0];
    const ggml_tensor* b = dst->src[1];
    const int dim        = dst->op_params[0];
    const int64_t n      = ggml_abi_nelements(dst);
    const size_t ts      = ggml_abi_type_size(dst->type);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int64_t idx[4] = {i % dst->ne[0], (i / dst->ne[0]) % dst->ne[1], (i / (dst->ne[0] * dst->ne[1])) % dst->ne[2], i / (dst->ne[0] * dst->ne[1] * dst->ne[2])};
        char* dp = mptr(dst) + idx[0] * dst->nb[0] + idx[1] * dst->nb[1] + idx[2] * dst->nb[2] + idx[3] * dst->nb[3];
        const ggml_tensor* s = a;
        if (idx[dim] >= a->ne[dim]) {
            s = b;
            idx[dim] -= a->ne[dim];
        }
        memcpy(dp, cptr(s) + idx[0] * s->nb[0] + idx[1] * s->nb[1] + idx[2] * s->nb[2] + idx[3] * s->nb[3], ts);
    }
}

void op_repeat(ggml_tensor* dst) {
    const ggml_tensor* s = dst->src[0];
    const int64_t n      = ggml_abi_nelements(dst);
    const size_t ts      = ggml_abi_type_size(dst->type);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t i0 = i % dst->ne[0], i1 = (i / dst->ne[0]) % dst->ne[1], i2 = (i / (dst->ne[0] * dst->ne[1])) % dst->ne[2], i3 = i / (dst->ne[0] * dst->ne[1] * dst->ne[2]);
        memcpy(mptr(dst) + i0 * dst->nb[0] + i1 * dst->nb[1] + i2 * dst->nb[2] + i3 * dst->nb[3],
               cptr(s) + (i0 % s->ne[0]) * s->nb[0] + (i1 % s->ne[1]) * s->nb[1] + (i2 % s->ne[2]) * s->nb[2] + (i3 % s->ne[3]) * s->nb[3], ts);
    }
}

// nearest: dst[i] = src[i / sf]  (block.hpp:61)
void op_upscale(ggml_tensor* dst) {
    const ggml_tensor* s = dst->src[0];
    const float sf0 = (float)dst->ne[0] / s->ne[0], sf1 = (float)dst->ne[1] / s->ne[1];
    const float sf2 = (float)dst->ne[2] / s->ne[2], sf3 = (float)dst->ne[3] / s->ne[3];
    const int64_t n = ggml_abi_nelements(dst);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t i0 = i % dst->ne[0], i1 = (i / dst->ne[0]) % dst->ne[1], i2 = (i / (dst->ne[0] * dst->ne[1])) % dst->ne[2], i3 = i / (dst->ne[0] * dst->ne[1] * dst->ne[2]);
        const int64_t j0 = (int64_t)(i0 / sf0), j1 = (int64_t)(i1 / sf1), j2 = (int64_t)(i2 / sf2), j3 = (int64_t)(i3 / sf3);
        *(float*)(mptr(dst) + i0 * dst->nb[0] + i1 * dst->nb[1] + i2 * dst->nb[2] + i3 * dst->nb[3]) =
            *(const float*)(cptr(s) + j0 * s->nb[0] + j1 * s->nb[1] + j2 * s->nb[2] + j3 * s->nb[3]);
    }
}

void op_pad(ggml_tensor* dst) {
    const ggml_tensor* s = dst->src[0];
    const int32_t* p     = dst->op_params;  // lp0,rp0,lp1,rp1,lp2,rp2,lp3,rp3
    const int64_t n      = ggml_abi_nelements(dst);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t i0 = i % dst->ne[0], i1 = (i / dst->ne[0]) % dst->ne[1], i2 = (i / (dst->ne[0] * dst->ne[1])) % dst->ne[2], i3 = i / (dst->ne[0] * dst->ne[1] * dst->ne[2]);
        const int64_t j0 = i0 - p[0], j1 = i1 - p[2], j2 = i2 - p[4], j3 = i3 - p[6];
        float v = 0.0f;
        if (j0 >= 0 && j0 < s->ne[0] && j1 >= 0 && j1 < s->ne[1] && j2 >= 0 && j2 < s->ne[2] && j3 >= 0 && j3 < s->ne[3])
            v = *(const float*)(cptr(s) + j0 * s->nb[0] + j1 * s->nb[1] + j2 * s->nb[2] + j3 * s->nb[3]);
        *(float*)(mptr(dst) + i0 * dst->nb[0] + i1 * dst->nb[1] + i2 * dst->nb[2] + i3 * dst->nb[3]) = v;
    }
}

// ggml_extend.hpp:1579-1606 / 1644-1652: cos first, then sin
void op_timestep_embedding(ggml_tensor* dst) {
    const ggml_tensor* t = dst->src[0];
    const int dim = dst->op_params[0], max_period = dst->op_params[1];
    const int half = dim / 2;
    for (int64_t i = 0; i < t->ne[0]; ++i) {
        float* emb        = (float*)(mptr(dst) + i * dst->nb[1]);
        const float tstep = ((const float*)cptr(t))[i];
        for (int j = 0; j < half; ++j) {
            const float freq = expf(-logf((float)max_period) * j / half);
            const float arg  = tstep * freq;
            emb[j]           = cosf(arg);
            emb[j + half]    = sinf(arg);
        }
        if (dim % 2 != 0) emb[2 * half] = 0.0f;
    }
}

void op_soft_max(ggml_tensor* dst) {
    const ggml_tensor* x    = dst->src[0];
    const ggml_tensor* mask = dst->src[1];
    const float scale       = opf(dst, 0);
    const int64_t nc = x->ne[0], nr = ggml_abi_nrows(x);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nr; ++r) {
        const int64_t i1 = r % x->ne[1], i2 = (r / x->ne[1]) % x->ne[2], i3 = r / (x->ne[1] * x->ne[2]);
        const float* xr = (const float*)(cptr(x) + i1 * x->nb[1] + i2 * x->nb[2] + i3 * x->nb[3]);
        float* yr       = (float*)(mptr(dst) + i1 * dst->nb[1] + i2 * dst->nb[2] + i3 * dst->nb[3]);
        const char* mr  = mask ? cptr(mask) + i1 * mask->nb[1] + (i2 % mask->ne[2]) * mask->nb[2] + (i3 % mask->ne[3]) * mask->nb[3] : nullptr;
        std::vector<float> wp(nc);
        float mx = -INFINITY;
        for (int64_t i = 0; i < nc; ++i) {
            float v = xr[i] * scale;
            if (mr) v += load_as_f32(mask, mr + i * mask->nb[0]);
            wp[i] = v;
            mx    = std::max(mx, v);
        }
        double sum = 0.0;
        for (int64_t i = 0; i < nc; ++i) {
            const float e = expf(wp[i] - mx);
            yr[i]         = e;
            sum += (double)e;
        }
        const float inv = (float)(1.0 / sum);
        for (int64_t i = 0; i < nc; ++i) yr[i] *= inv;
    }
}

// ggml-cpu flash_attn_ext (f16 K/V path): SURVEY.md Appendix A + E.3; call site ggml_extend.hpp:1424
void op_flash_attn_ext(ggml_tensor* dst) {
    const ggml_tensor* q = dst->src[0];
    const ggml_tensor* k = dst->src[1];
    const ggml_tensor* v = dst->src[2];
    const ggml_tensor* mask = dst->src[3];
    const float scale       = opf(dst, 0);
    const int64_t DK = k->ne[0], DV = v->ne[0], Lq = q->ne[1], Lk = k->ne[1], H = q->ne[2], B = q->ne[3];
    const int64_t rk2 = q->ne[2] / k->ne[2], rv2 = q->ne[2] / v->ne[2];
    const bool v_f16 = v->type == GGML_TYPE_F16;
#pragma omp parallel for collapse(3) schedule(static)
    for (int64_t ib = 0; ib < B; ++ib)
        for (int64_t ih = 0; ih < H; ++ih)
            for (int64_t iq = 0; iq < Lq; ++iq) {
                const float* pq = (const float*)(cptr(q) + iq * q->nb[1] + ih * q->nb[2] + ib * q->nb[3]);
                std::vector<float> Qh(DK);  // q row after the round trip through K's vec_dot_type
                for (int64_t d = 0; d < DK; ++d) Qh[d] = (k->type == GGML_TYPE_F16) ? h2f(f2h(pq[d])) : pq[d];
                std::vector<float> VKQ32(DV, 0.0f);
                std::vector<ggml_fp16_t> VKQ16(DV, 0);
                float S = 0.0f, M = -INFINITY;
                const char* mp = mask ? cptr(mask) + iq * mask->nb[1] + (ih % mask->ne[2]) * mask->nb[2] + (ib % mask->ne[3]) * mask->nb[3] : nullptr;
                for (int64_t ic = 0; ic < Lk; ++ic) {
                    const float mv = mp ? load_as_f32(mask, mp + ic * mask->nb[0]) : 0.0f;
                    if (mv == -INFINITY) continue;
                    const char* kr = cptr(k) + ic * k->nb[1] + (ih / rk2) * k->nb[2] + ib * k->nb[3];
                    float s        = 0.0f;
                    for (int64_t d = 0; d < DK; ++d) s += load_as_f32(k, kr + d * k->nb[0]) * Qh[d];
                    s = s * scale + mv;
                    const float Mold = M;
                    float ms = 1.0f, vs = 1.0f;
                    const char* vr = cptr(v) + ic * v->nb[1] + (ih / rv2) * v->nb[2] + ib * v->nb[3];
                    if (s > M) {
                        M  = s;
                        ms = expf(Mold - M);
                        if (v_f16)
                            for (int64_t d = 0; d < DV; ++d) VKQ16[d] = f2h(h2f(VKQ16[d]) * ms);
                        else
                            for (int64_t d = 0; d < DV; ++d) VKQ32[d] *= ms;
                    } else {
                        vs = expf(s - M);
                    }
                    if (v_f16)
                        for (int64_t d = 0; d < DV; ++d) VKQ16[d] = f2h(h2f(VKQ16[d]) + h2f(((const ggml_fp16_t*)vr)[d]) * vs);
                    else
                        for (int64_t d = 0; d < DV; ++d) VKQ32[d] += load_as_f32(v, vr + d * v->nb[0]) * vs;
                    S = S * ms + vs;
                }
                if (v_f16)
                    for (int64_t d = 0; d < DV; ++d) VKQ32[d] = h2f(VKQ16[d]);
                const float Sinv = S == 0.0f ? 0.0f : 1.0f / S;
                // dst.ne = [DV, H, Lq, B]
                float* out = (float*)(mptr(dst) + ih * dst->nb[1] + iq * dst->nb[2] + ib * dst->nb[3]);
                for (int64_t d = 0; d < DV; ++d) out[d] = VKQ32[d] * Sinv;
            }
}

void op_get_rows(ggml_tensor* dst) {
    const ggml_tensor* a = dst->src[0];
    const ggml_tensor* b = dst->src[1];
    const int64_t nc = a->ne[0], nr = ggml_abi_nelements(b);
    for (int64_t i = 0; i < nr; ++i) {
        const int64_t i10 = i % b->ne[0], i11 = (i / b->ne[0]) % b->ne[1], i12 = i / (b->ne[0] * b->ne[1]);
        const int32_t row = *(const int32_t*)(cptr(b) + i10 * b->nb[0] + i11 * b->nb[1] + i12 * b->nb[2]);
        const char* src   = cptr(a) + row * a->nb[1] + i11 * a->nb[2] + i12 * a->nb[3];
        float* out        = (float*)(mptr(dst) + i10 * dst->nb[1] + i11 * dst->nb[2] + i12 * dst->nb[3]);
        if (a->type == GGML_TYPE_Q8_0) {  // block_q8_0: f16 d, 32 x int8; value = d * q
            for (int64_t c = 0; c < nc; ++c) {
                const char* blk = src + (c / 32) * 34;
                out[c]          = h2f(*(const ggml_fp16_t*)blk) * (float)((const int8_t*)(blk + 2))[c % 32];
            }
        } else if (a->type == GGML_TYPE_Q4_0) {  // block_q4_0: f16 d, 16 bytes; element j < 16 = low nibble of qs[j], j >= 16 = high nibble of qs[j-16]
            for (int64_t c = 0; c < nc; ++c) {
                const char* blk = src + (c / 32) * 18;
                const int j     = (int)(c % 32);
                const uint8_t q = ((const uint8_t*)(blk + 2))[j % 16];
                out[c]          = h2f(*(const ggml_fp16_t*)blk) * (float)((j < 16 ? (q & 0xF) : (q >> 4)) - 8);
            }
        } else {
            for (int64_t c = 0; c < nc; ++c) out[c] = load_as_f32(a, src + c * a->nb[0]);
        }
    }
}

bool supported(const ggml_tensor* t) {
    switch (t->op) {
        case GGML_OP_NONE:
        case GGML_OP_RESHAPE:
        case GGML_OP_VIEW:
        case GGML_OP_PERMUTE:
        case GGML_OP_TRANSPOSE:
        case GGML_OP_DUP:
        case GGML_OP_CONT:
        case GGML_OP_CPY:
        case GGML_OP_ADD:
        case GGML_OP_SUB:
        case GGML_OP_MUL:
        case GGML_OP_DIV:
        case GGML_OP_SCALE:
        case GGML_OP_NORM:
        case GGML_OP_RMS_NORM:
        case GGML_OP_GROUP_NORM:
        case GGML_OP_MUL_MAT:
        case GGML_OP_IM2COL:
        case GGML_OP_CONV_2D:
        case GGML_OP_CONCAT:
        case GGML_OP_REPEAT:
        case GGML_OP_UPSCALE:
        case GGML_OP_PAD:
        case GGML_OP_TIMESTEP_EMBEDDING:
        case GGML_OP_SOFT_MAX:
        case GGML_OP_FLASH_ATTN_EXT:
        case GGML_OP_GET_ROWS:
            return true;
        case GGML_OP_UNARY:
            switch (ggml_abi_get_unary_op(t)) {
                case GGML_UNARY_OP_SILU:
                case GGML_UNARY_OP_SIGMOID:
                case GGML_UNARY_OP_TANH:
                case GGML_UNARY_OP_RELU:
                case GGML_UNARY_OP_GELU:
                case GGML_UNARY_OP_GELU_QUICK:
                case GGML_UNARY_OP_NEG:
                case GGML_UNARY_OP_EXP:
                    return true;
                default:
                    return false;
            }
        default:
            return false;
    }
}

enum ggml_status compute_node(ggml_tensor* n) {
    switch (n->op) {
        case GGML_OP_NONE:
        case GGML_OP_RESHAPE:
        case GGML_OP_VIEW:
        case GGML_OP_PERMUTE:
        case GGML_OP_TRANSPOSE:
            return GGML_STATUS_SUCCESS;
        case GGML_OP_DUP:
        case GGML_OP_CONT:
        case GGML_OP_CPY: op_copy(n); break;
        case GGML_OP_ADD:
        case GGML_OP_SUB:
        case GGML_OP_MUL:
        case GGML_OP_DIV: op_binary(n); break;
        case GGML_OP_SCALE: op_scale(n); break;
        case GGML_OP_NORM: op_norm(n, false); break;
        case GGML_OP_RMS_NORM: op_norm(n, true); break;
        case GGML_OP_GROUP_NORM: op_group_norm(n); break;
        case GGML_OP_MUL_MAT: op_mul_mat(n); break;
        case GGML_OP_IM2COL: op_im2col(n); break;
        case GGML_OP_CONV_2D: op_conv_2d(n); break;
        case GGML_OP_CONCAT: op_concat(n); break;
        case GGML_OP_REPEAT: op_repeat(n); break;
        case GGML_OP_UPSCALE: op_upscale(n); break;
        case GGML_OP_PAD: op_pad(n); break;
        case GGML_OP_TIMESTEP_EMBEDDING: op_timestep_embedding(n); break;
        case GGML_OP_SOFT_MAX: op_soft_max(n); break;
        case GGML_OP_FLASH_ATTN_EXT: op_flash_attn_ext(n); break;
        case GGML_OP_GET_ROWS: op_get_rows(n); break;
        case GGML_OP_UNARY: op_unary(n); break;
        default:
            fprintf(stderr, "cpu-oracle: unsupported op %d (%s)\n", (int)n->op, n->name);
            return GGML_STATUS_FAILED;
    }
    return GGML_STATUS_SUCCESS;
}

// ---------------------------------------------------------------- backend plumbing (host memory)
ggml_backend_reg g_reg;
ggml_backend_device g_dev;
ggml_backend_buffer_type g_buft;
ggml_guid g_guid = {{0x0a, 0xc1, 0xe0, 0x51, 0x0d, 0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0xaa, 0xbb}};

const char* buft_name(ggml_backend_buffer_type_t) { return "CPU-oracle"; }
void buf_free(ggml_backend_buffer_t b) { free(b->context); }
void* buf_base(ggml_backend_buffer_t b) { return b->context; }
void buf_memset(ggml_backend_buffer_t, ggml_tensor* t, uint8_t v, size_t off, size_t sz) { memset((char*)t->data + off, v, sz); }
void buf_set(ggml_backend_buffer_t, ggml_tensor* t, const void* d, size_t off, size_t sz) { memcpy((char*)t->data + off, d, sz); }
void buf_get(ggml_backend_buffer_t, const ggml_tensor* t, void* d, size_t off, size_t sz) { memcpy(d, (const char*)t->data + off, sz); }
void buf_clear(ggml_backend_buffer_t b, uint8_t v) { memset(b->context, v, b->size); }
ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, std::max<size_t>(size, 64)) != 0) return nullptr;
    ggml_backend_buffer* b = new ggml_backend_buffer();
    memset(b, 0, sizeof(*b));
    b->iface.free_buffer   = buf_free;
    b->iface.get_base      = buf_base;
    b->iface.memset_tensor = buf_memset;
    b->iface.set_tensor    = buf_set;
    b->iface.get_tensor    = buf_get;
    b->iface.clear         = buf_clear;
    b->buft                = buft;
    b->context             = p;
    b->size                = size;
    b->usage               = GGML_BACKEND_BUFFER_USAGE_ANY;
    return b;
}
size_t buft_align(ggml_backend_buffer_type_t) { return 64; }
size_t buft_max(ggml_backend_buffer_type_t) { return SIZE_MAX; }
size_t buft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor* t) { return ggml_abi_nbytes(t); }
bool buft_is_host(ggml_backend_buffer_type_t) { return true; }

const char* be_name(ggml_backend_t) { return "CPU-oracle"; }
void be_free(ggml_backend_t b) { delete b; }
void be_sync(ggml_backend_t) {}
enum ggml_status be_graph_compute(ggml_backend_t, ggml_cgraph* g) {
    init_tables();
    static const bool prof = getenv("ORACLE_PROFILE") != nullptr;  // per-op wall time of every graph, printed to stderr (where does an oracle forward spend its time?)
    double t_op[GGML_OP_COUNT + 1] = {0};
    for (int i = 0; i < g->n_nodes; ++i) {
        const double t0 = prof ? omp_get_wtime() : 0.0;
        enum ggml_status st = compute_node(g->nodes[i]);
        if (st != GGML_STATUS_SUCCESS) return st;
        if (prof) t_op[std::min<int>((int)g->nodes[i]->op, GGML_OP_COUNT)] += omp_get_wtime() - t0;
    }
    if (prof) {
        double tot = 0;
        for (double v : t_op) tot += v;
        fprintf(stderr, "[oracle profile] %d nodes, %.3f s:", g->n_nodes, tot);
        for (int o = 0; o <= GGML_OP_COUNT; ++o)
            if (t_op[o] > 0.01 * tot) fprintf(stderr, " op%d %.3f", o, t_op[o]);
        fprintf(stderr, "\n");
    }
    return GGML_STATUS_SUCCESS;
}

const char* dev_name(ggml_backend_dev_t) { return "CPU-oracle"; }
const char* dev_desc(ggml_backend_dev_t) { return "CPU restatement of the reference ggml-cpu path (test oracle)"; }
void dev_memory(ggml_backend_dev_t, size_t* free, size_t* total) { *free = *total = (size_t)1 << 36; }
enum ggml_backend_dev_type dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_CPU; }
void dev_props(ggml_backend_dev_t d, ggml_backend_dev_props* p) {
    memset(p, 0, sizeof(*p));
    p->name        = dev_name(d);
    p->description = dev_desc(d);
    dev_memory(d, &p->memory_free, &p->memory_total);
    p->type             = GGML_BACKEND_DEVICE_TYPE_CPU;
    p->caps.host_buffer = true;
}
ggml_backend_t dev_init(ggml_backend_dev_t d, const char*) {
    ggml_backend* b = new ggml_backend();
    memset(b, 0, sizeof(*b));
    b->guid                = &g_guid;
    b->iface.get_name      = be_name;
    b->iface.free          = be_free;
    b->iface.synchronize   = be_sync;
    b->iface.graph_compute = be_graph_compute;
    b->device              = d;
    return b;
}
ggml_backend_buffer_type_t dev_buft(ggml_backend_dev_t) { return &g_buft; }
bool dev_supports_op(ggml_backend_dev_t, const ggml_tensor* op) { return supported(op); }
bool dev_supports_buft(ggml_backend_dev_t, ggml_backend_buffer_type_t b) { return b == &g_buft; }

const char* reg_name(ggml_backend_reg_t) { return "CPU-oracle"; }
size_t reg_count(ggml_backend_reg_t) { return 1; }
ggml_backend_dev_t reg_dev(ggml_backend_reg_t, size_t) { return &g_dev; }
void* reg_proc(ggml_backend_reg_t, const char*) { return nullptr; }

}  // namespace

extern "C" __attribute__((visibility("default"))) ggml_backend_reg_t ggml_backend_init(void) {
    static bool once = false;
    if (!once) {
        once = true;
        init_tables();
        memset(&g_buft, 0, sizeof(g_buft));
        g_buft.iface.get_name       = buft_name;
        g_buft.iface.alloc_buffer   = buft_alloc;
        g_buft.iface.get_alignment  = buft_align;
        g_buft.iface.get_max_size   = buft_max;
        g_buft.iface.get_alloc_size = buft_alloc_size;
        g_buft.iface.is_host        = buft_is_host;
        g_buft.device               = &g_dev;
        memset(&g_dev, 0, sizeof(g_dev));
        g_dev.iface.get_name        = dev_name;
        g_dev.iface.get_description = dev_desc;
        g_dev.iface.get_memory      = dev_memory;
        g_dev.iface.get_type        = dev_type;
        g_dev.iface.get_props       = dev_props;
        g_dev.iface.init_backend    = dev_init;
        g_dev.iface.get_buffer_type = dev_buft;
        g_dev.iface.supports_op     = dev_supports_op;
        g_dev.iface.supports_buft   = dev_supports_buft;
        g_dev.reg                   = &g_reg;
        memset(&g_reg, 0, sizeof(g_reg));
        g_reg.api_version            = GGML_BACKEND_API_VERSION;
        g_reg.iface.get_name         = reg_name;
        g_reg.iface.get_device_count = reg_count;
        g_reg.iface.get_device       = reg_dev;
        g_reg.iface.get_proc_address = reg_proc;
    }
    return &g_reg;
}
extern "C" __attribute__((visibility("default"))) int ggml_backend_score(void) { return 1; }
// number of OpenMP threads the oracle will use (reported as cpu_baseline.cores)
extern "C" __attribute__((visibility("default"))) int oracle_num_threads(void) { return omp_get_max_threads(); }
// team size for the following graph computes: n > 0 = exactly n, n <= 0 = every schedulable CPU (affinity mask and cgroup quota; hardware threads,
// not physical cores).  Returns the size set.  bench.py's cpu_baseline reports the default (<= 16) and the all-CPU figure; the full-depth DiT parity
// tests use it so the oracle forward finishes in minutes.
extern "C" __attribute__((visibility("default"))) int oracle_set_num_threads(int n) {
    init_tables();
    const int want = n > 0 ? n : usable_cpus(false);
    omp_set_num_threads(want);
    return want;
}
// 1: every MUL_MAT with a 16-bit or quantised weight multiplies the exactly widened weight with the UNROUNDED f32 activations (no f16 / bf16 / q8_0
// rounding of src1): the arithmetic-exact reference the full-depth DiT tests measure both the ggml-cpu-faithful path and the GPU against.  Returns the old value.
extern "C" __attribute__((visibility("default"))) int oracle_set_exact_weights(int on) {
    const int was   = g_exact_weights ? 1 : 0;
    g_exact_weights = on != 0;
    return was;
}
