// ktime.hip — see ktime.h
#include "ktime.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

namespace mi355x {
namespace {
const char* const kNames[KF_COUNT] = {
    "conv implicit-GEMM, 256-row tiles (k_conv3w<*,*> LDS-window kernel / k_gemm16<256,*,true,...>)",
    "conv implicit-GEMM, 128-row tiles (k_gemm16<128,*,true,...>)",
    "Linear MFMA GEMM (k_gemm16<*,*,false,...>)",
    "flash attention (k_flash_attn)",
    "weight-streaming few-row Linears (k_qgemv / k_qgemm16: raw q8_0 / q4_0 blocks, in-register dequant; k_fgemv: f16 / f32)",
    "f32 MFMA matmul (k_mul_mat_generic)",
    "GroupNorm apply + SiLU + NCHW->NHWC f16 (k_nchw_to_nhwc_f16)",
    "LayerNorm -> f16 operand image (k_layer_norm_f16)",
    "GroupNorm statistics (k_gn_stats)",
    "f32 rows -> f16 operand image (k_pack_rows_f16 / k_geglu_f16)",
    "copies / transposes (k_copy_*, k_transpose)",
    "binary elementwise (k_bin_*)",
    "concat (k_concat*)",
    "unary / scale (k_unary, k_scale)",
    "split-K slab reduce (k_splitk_reduce)",
    "f32 norms (k_group_norm, k_layer_norm)",
    "softmax (k_soft_max)",
    "other",
};
const int kBound[KF_COUNT] = {0, 0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

struct Rec {
    hipEvent_t e0, e1;
    int fam;
    double flops, bytes;
};
std::atomic<uint32_t> g_mask{0};
std::mutex g_mu;
// events belong to the device that was current when they were created: pool and records are kept PER DEVICE (several backend instances may live in
// one process, shard.generate_multi_device), and the record list is capped (nobody may be reading while timing stays enabled)
constexpr int KT_MAX_DEV   = 16;
constexpr size_t KT_MAX_REC = 1u << 20;
struct DevState {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    std::vector<Rec> recs;
};
DevState g_dev[KT_MAX_DEV];
}  // namespace

void ktime_enable(uint32_t fam_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < KT_MAX_DEV; ++d) {
        if (g_dev[d].recs.empty()) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        g_dev[d].recs.clear();
    }
    (void)hipSetDevice(cur);
    (void)hipDeviceSynchronize();
    g_mask.store(fam_mask, std::memory_order_relaxed);
}
bool ktime_any() { return g_mask.load(std::memory_order_relaxed) != 0; }
bool ktime_on(int fam) { return (g_mask.load(std::memory_order_relaxed) >> fam) & 1u; }

KScope::KScope(hipStream_t stream, int fam, double flops, double bytes) : s(stream) {
    if (!ktime_on(fam)) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KT_MAX_DEV) return;
    std::lock_guard<std::mutex> lk(g_mu);
    DevState& D = g_dev[dev];
    if (D.recs.size() >= KT_MAX_REC) return;  // full: further launches go untimed until the next read
    if (D.recs.size() == D.pool.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        D.pool.emplace_back(a, b);
    }
    const auto& ev = D.pool[D.recs.size()];
    D.recs.push_back({ev.first, ev.second, fam, flops, bytes});
    (void)hipEventRecord(ev.first, s);
    e1 = ev.second;
}

int ktime_read(KFamTiming* out, int cap, int* fam_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    struct Done {
        int fam;
        double flops, bytes;
        float ms;
    };
    std::vector<Done> g_recs;  // the finished launches of every device (elapsed times are read while the events' device is current)
    for (int d = 0; d < KT_MAX_DEV; ++d) {
        if (g_dev[d].recs.empty()) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        for (const Rec& r : g_dev[d].recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) g_recs.push_back({r.fam, r.flops, r.bytes, ms});
        }
        g_dev[d].recs.clear();
    }
    (void)hipSetDevice(cur);
    (void)hipDeviceSynchronize();
    KFamTiming acc[KF_COUNT];
    for (int f = 0; f < KF_COUNT; ++f) acc[f] = {kNames[f], kBound[f], 0, 0.0, 0.0, 0.0};
    for (const Done& r : g_recs) {
        acc[r.fam].launches++;
        acc[r.fam].total_ms += r.ms;
        acc[r.fam].total_flops += r.flops;
        acc[r.fam].total_bytes += r.bytes;
    }
    // MI355X_KTIME_DUMP=<file>: append one line per distinct (family, flops, bytes) launch shape — which shapes a family's time sits in
    if (const char* path = getenv("MI355X_KTIME_DUMP")) {
        std::map<std::tuple<int, double, double>, std::pair<int64_t, double>> shapes;
        for (const Done& r : g_recs) {
            auto& a = shapes[{r.fam, r.flops, r.bytes}];
            a.first++;
            a.second += r.ms;
        }
        if (FILE* fp = fopen(path, "a")) {
            fprintf(fp, "# family | launches | us/launch | total ms | GFLOP/launch | MB/launch | TFLOP/s | GB/s\n");
            for (const auto& kv : shapes) {
                const double us = kv.second.second * 1e3 / (double)kv.second.first;
                fprintf(fp, "%-28.28s | %5lld | %9.1f | %8.3f | %10.3f | %9.2f | %7.1f | %7.1f\n", kNames[std::get<0>(kv.first)],
                        (long long)kv.second.first, us, kv.second.second, std::get<1>(kv.first) * 1e-9, std::get<2>(kv.first) * 1e-6,
                        std::get<1>(kv.first) / us * 1e-6, std::get<2>(kv.first) / us * 1e-3);
            }
            fclose(fp);
        }
    }
    int n = 0;
    for (int f = 0; f < KF_COUNT && n < cap; ++f)
        if (acc[f].launches > 0) {
            if (fam_index) fam_index[n] = f;
            out[n++] = acc[f];
        }
    return n;
}

}  // namespace mi355x
