// ktime.h — live per-kernel-family timing for bench.py's roofline legs: while a family's bit is enabled, every dispatch of that family is
// bracketed by two HIP events recorded on the launch stream; the launch's algorithmic FLOPs and HBM bytes (one
// read + one write of the activation for bandwidth kernels, 2*M*N*K for contractions) are accumulated beside the durations.
// Off (mask 0) costs one relaxed load per launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355x {

enum KFam {
    KF_CONV_T256 = 0,   // implicit-GEMM conv, 256-row tiles (k_gemm16<256, ...>)
    KF_CONV_T128,       // implicit-GEMM conv, 128-row tiles
    KF_LINEAR,          // Linear / MUL_MAT with a static weight on the MFMA GEMM (all tiles, incl. GEGLU / head-major / gated epilogues)
    KF_FLASH,           // k_flash_attn
    KF_QGEMM,           // in-register q8_0 / q4_0 dequant GEMM (raw quantised blocks streamed from HBM)
    KF_GEMM_F32,        // k_mul_mat_generic (exact f32 MFMA)
    KF_NCHW_NHWC,       // k_nchw_to_nhwc_f16 (GroupNorm apply + SiLU + layout -> conv operand image)
    KF_LN_F16,          // k_layer_norm_f16 (LayerNorm / RMSNorm / adaLN modulate -> Linear operand image)
    KF_GN_STATS,        // k_gn_stats
    KF_PACK_F16,        // k_pack_rows_f16 / k_geglu_f16
    KF_COPY,            // k_copy_* / k_transpose (CONT, CPY, permutes)
    KF_BINARY,          // k_bin_*
    KF_CONCAT,          // k_concat*
    KF_UNARY,           // k_unary / k_scale
    KF_SPLITK,          // k_splitk_reduce
    KF_NORM_F32,        // k_group_norm / k_layer_norm (f32 outputs no fusion claimed)
    KF_SOFTMAX,
    KF_OTHER,
    KF_COUNT
};

struct KFamTiming {
    const char* name;
    int bound;  // 0 = MFMA (FLOP/s), 1 = HBM (B/s)
    int64_t launches;
    double total_ms, total_flops, total_bytes;
};

void ktime_enable(uint32_t fam_mask);  // resets the accumulators
bool ktime_on(int fam);
bool ktime_any();  // any family enabled: plans run eagerly (events cannot be read back from inside a captured hipGraph)
int ktime_read(KFamTiming* out, int cap, int* fam_index = nullptr);  // synchronises the device; returns the families with launches > 0; resets

struct KScope {
    hipStream_t s;
    hipEvent_t e1 = nullptr;
    KScope(hipStream_t stream, int fam, double flops, double bytes);
    ~KScope() {
        if (e1) (void)hipEventRecord(e1, s);
    }
    KScope(const KScope&)            = delete;
    KScope& operator=(const KScope&) = delete;
};

}  // namespace mi355x
