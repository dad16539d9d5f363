// engine.cpp — host driver of the hot path + the C ABI declared in include/sd-mi355x.h.
//
// Mirrors (re-designed, not transcribed):
//   GGMLRunner::compute / execute_graph      src/core/ggml_extend.hpp:3151-3210, 2767-2930
//   StableDiffusionGGML::sample + denoise    src/stable-diffusion.cpp:2509-2926
//   sample_euler_ancestral / sample_euler    src/runtime/denoiser.hpp:1513-1546, 1582-1597
//   ClassifierFreeGuidance::forward          src/runtime/guidance.cpp:149-179
//   decode_first_stage / VAE::decode         src/stable-diffusion.cpp:3062-3078, src/model/vae/vae.hpp:170-222
//   sdm_generate_image batch loop                src/stable-diffusion.cpp:5664-5721
//
// Like the reference, the graph is rebuilt for every model call; unlike it, `device_batch`
// images are denoised together in ONE graph (N>1 is our extension, F6) — per-image results are defined
// as those of independent batch-1 runs with seeds seed+b.
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "ggml.h"
#include "models.hpp"
#include "conditioner.hpp"
#include "sampler.hpp"
#include "model_io.hpp"
#include "torch_ckpt_io.hpp"
#include "name_conversion.hpp"
#include "sd-mi355x.h"

using namespace sdmi;

static thread_local std::string g_last_error;
static void set_error(const std::string& e) {
    g_last_error = e;
    fprintf(stderr, "[sd-mi355x] error: %s\n", e.c_str());
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// host post-processing of whole image batches (clamp / uint8 conversion of 6.3 M floats for 8 x 512x512) on a few threads
template <typename Fn>
static void parallel_chunks(size_t n, Fn&& fn, size_t min_parallel = (1u << 16)) {
    const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const size_t T    = n < min_parallel ? 1 : std::min<size_t>(hw, n);
    if (T == 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t step = (n + T - 1) / T;
    for (size_t t = 0; t < T; ++t) {
        const size_t b = t * step, e = std::min(n, b + step);
        if (b < e) th.emplace_back([&fn, b, e]() { fn(b, e); });
    }
    for (auto& t : th) t.join();
}

// ---------------------------------------------------------------------------------------------------
// synthetic weights: deterministic per tensor name, N(0, 1/sqrt(fan_in)) weights, small biases,
// norm scales around 1
// ---------------------------------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t hash_name(const std::string& s, uint64_t seed) {
    uint64_t h = 1469598103934665603ull ^ seed;
    for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
    return h;
}
static void fill_normal(float* dst, int64_t n, uint64_t seed, float mean, float std) {
    const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(std::thread::hardware_concurrency(), 16u), n / 65536));
    auto work          = [&](int tid) {
        const int64_t chunk = (n + nthreads - 1) / nthreads;
        const int64_t i0 = tid * chunk, i1 = std::min<int64_t>(n, i0 + chunk);
        // counter-based: value i depends only on (seed, i/2) so the result is thread-count independent
        for (int64_t i = i0 & ~1ll; i < i1; i += 2) {
            uint64_t s  = seed + (uint64_t)(i / 2) * 0x632BE59BD9B4E019ull;
            uint64_t r1 = splitmix64(s), r2 = splitmix64(s);
            const float u1 = ((float)(r1 >> 40) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = ((float)(r2 >> 40) + 0.5f) * (1.0f / 16777216.0f);
            const float m  = sqrtf(-2.0f * logf(u1));
            const float a = m * cosf(6.2831853f * u2), b = m * sinf(6.2831853f * u2);
            if (i >= i0 && i < i1) dst[i] = mean + std * a;
            if (i + 1 >= i0 && i + 1 < i1) dst[i + 1] = mean + std * b;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
}

// ---------------------------------------------------------------------------------------------------
// node-by-node evaluation hook (include/sd-mi355x.h): restates src/core/ggml_extend_backend.cpp:449-509 and src/core/util.cpp:638-668
// ---------------------------------------------------------------------------------------------------
static sdm_graph_eval_callback_t g_eval_cb = nullptr;
static void* g_eval_cb_data                = nullptr;

static ggml_cgraph graph_view(ggml_cgraph* parent, int i0, int i1) {  // sd_ggml_graph_view
    ggml_cgraph v;
    v.size             = 0;
    v.n_nodes          = i1 - i0;
    v.n_leafs          = 0;
    v.nodes            = parent->nodes + i0;
    v.grads            = nullptr;
    v.grad_accs        = nullptr;
    v.leafs            = nullptr;
    v.use_counts       = parent->use_counts;
    v.visited_hash_set = parent->visited_hash_set;
    v.order            = parent->order;
    v.uid              = 0;
    return v;
}

extern "C" void sdm_set_backend_eval_callback(sdm_graph_eval_callback_t cb, void* user_data) {
    g_eval_cb      = cb;
    g_eval_cb_data = user_data;
}

extern "C" int sdm_backend_graph_compute_with_eval_callback(ggml_backend* backend, ggml_cgraph* gf, sdm_graph_eval_callback_t cb, void* user_data) {
    if (cb == nullptr) return (int)ggml_backend_graph_compute(backend, gf);
    enum ggml_status status = GGML_STATUS_SUCCESS;
    const int n_nodes       = ggml_graph_n_nodes(gf);
    bool stopped            = false;
    for (int j0 = 0; j0 < n_nodes; ++j0) {
        ggml_tensor* t = ggml_graph_node(gf, j0);
        bool need      = cb(t, true, user_data);
        int j1         = j0;
        while (!need && j1 < n_nodes - 1) {
            t    = ggml_graph_node(gf, ++j1);
            need = cb(t, true, user_data);
        }
        ggml_cgraph gv = graph_view(gf, j0, j1 + 1);
        status         = ggml_backend_graph_compute_async(backend, &gv);
        if (status != GGML_STATUS_SUCCESS) break;
        ggml_backend_synchronize(backend);
        if (need && !cb(t, false, user_data)) {
            stopped = true;
            break;
        }
        j0 = j1;
    }
    ggml_backend_synchronize(backend);
    if (stopped && status == GGML_STATUS_SUCCESS) status = GGML_STATUS_ABORTED;
    return (int)status;
}

// ---------------------------------------------------------------------------------------------------
// graph topology as text (include/sd-mi355x.h: sdm_graph_describe)
// ---------------------------------------------------------------------------------------------------
static std::string describe_graph(ggml_cgraph* gf) {
    std::unordered_map<const ggml_tensor*, std::string> ref;
    std::vector<const ggml_tensor*> leafs;
    auto leaf_ref = [&](const ggml_tensor* t) -> const std::string& {
        auto it = ref.find(t);
        if (it != ref.end()) return it->second;
        leafs.push_back(t);
        return ref[t] = "l" + std::to_string(leafs.size() - 1);
    };
    for (int i = 0; i < gf->n_nodes; ++i) ref[gf->nodes[i]] = "n" + std::to_string(i);
    if (gf->leafs)
        for (int j = 0; j < gf->n_leafs; ++j) leaf_ref(gf->leafs[j]);
    std::string nodes;
    char line[1024];
    for (int i = 0; i < gf->n_nodes; ++i) {
        const ggml_tensor* t = gf->nodes[i];
        int n = snprintf(line, sizeof(line), "N %d %s %s ne=%lld,%lld,%lld,%lld nb=%zu,%zu,%zu,%zu p=", i, ggml_op_name(t->op), ggml_type_name(t->type), (long long)t->ne[0],
                         (long long)t->ne[1], (long long)t->ne[2], (long long)t->ne[3], t->nb[0], t->nb[1], t->nb[2], t->nb[3]);
        nodes.append(line, n);
        int last = -1;
        for (int k = 0; k < (int)(sizeof(t->op_params) / sizeof(int32_t)); ++k)
            if (t->op_params[k] != 0) last = k;
        for (int k = 0; k <= last; ++k) {
            n = snprintf(line, sizeof(line), k ? ",%d" : "%d", t->op_params[k]);
            nodes.append(line, n);
        }
        n = snprintf(line, sizeof(line), " f=%d s=", t->flags);
        nodes.append(line, n);
        bool any = false;
        for (int k = 0; k < GGML_MAX_SRC; ++k) {
            if (!t->src[k]) continue;
            if (any) nodes += ',';
            nodes += std::to_string(k) + ":" + leaf_ref(t->src[k]);
            any = true;
        }
        if (!any) nodes += '-';
        nodes += " v=";
        if (t->view_src) {
            nodes += leaf_ref(t->view_src) + "@" + std::to_string(t->view_offs);
        } else {
            nodes += '-';
        }
        nodes += " name=";
        nodes += t->name;
        nodes += '\n';
    }
    std::string out;
    for (size_t j = 0; j < leafs.size(); ++j) {
        const ggml_tensor* t = leafs[j];
        const int n = snprintf(line, sizeof(line), "L %zu %s ne=%lld,%lld,%lld,%lld nb=%zu,%zu,%zu,%zu f=%d name=%s\n", j, ggml_type_name(t->type), (long long)t->ne[0], (long long)t->ne[1],
                               (long long)t->ne[2], (long long)t->ne[3], t->nb[0], t->nb[1], t->nb[2], t->nb[3], t->flags, t->name);
        out.append(line, n);
    }
    return out + nodes;
}
static int g_graph_capture = 0;  // 1: describe every graph before it is submitted; 2: describe it and do NOT compute (topology of full-size models on a CPU box)
static std::string g_last_graph;
extern "C" size_t sdm_graph_describe(ggml_cgraph* gf, char* buf, size_t cap) {
    const std::string d = describe_graph(gf);
    if (buf && cap > 0) {
        const size_t n = std::min(cap - 1, d.size());
        memcpy(buf, d.data(), n);
        buf[n] = 0;
    }
    return d.size() + 1;
}
extern "C" void sdm_set_graph_capture(int on) {
    g_graph_capture = on;
    if (!on) g_last_graph.clear();
}
extern "C" size_t sdm_last_graph_description(char* buf, size_t cap) {
    if (buf && cap > 0) {
        const size_t n = std::min(cap - 1, g_last_graph.size());
        memcpy(buf, g_last_graph.data(), n);
        buf[n] = 0;
    }
    return g_last_graph.size() + 1;
}

// ---------------------------------------------------------------------------------------------------
// Runner: weights residency + per-call graph lifecycle
// ---------------------------------------------------------------------------------------------------
struct HostInput {
    ggml_tensor* t;
    const void* data;
    size_t nbytes;
};

struct Runner {
    ggml_backend_t backend        = nullptr;
    ParamStore ps;
    ggml_backend_buffer_t weights = nullptr;
    ggml_gallocr_t galloc         = nullptr;
    size_t graph_size             = 102400;  // UNET_GRAPH_SIZE (unet.hpp:14)
    int64_t calls                 = 0;
    int64_t last_nodes            = 0;
    double build_ms = 0, alloc_ms = 0, submit_ms = 0;  // cumulative host time: graph construction, gallocr, uploads + graph_compute call

    ~Runner() {
        drop_cache();
        if (galloc) ggml_gallocr_free(galloc);
        if (weights) ggml_backend_buffer_free(weights);
    }

    bool alloc_weights(uint64_t seed) {
        weights = ggml_backend_alloc_ctx_tensors(ps.ctx, backend);
        if (!weights) return false;
        ggml_backend_buffer_set_usage(weights, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
        // graph-topology tests of the full-size models (tests/test_ref_graphs.py) build and describe graphs without computing them: the 8-12 B parameter
        // tables are allocated (untouched pages) and left unfilled
        if (getenv("SDCPP_SKIP_WEIGHT_INIT")) return true;
        std::vector<float> tmp;
        std::vector<uint8_t> conv;
        double t_fill = 0, t_up = 0;
        for (auto& sp : ps.specs) {
            const int64_t n = ggml_nelements(sp.tensor);
            if ((int64_t)tmp.size() < n) tmp.resize(n);  // grow only: every element is overwritten below
            const uint64_t s = hash_name(sp.name, seed);
            const double t0  = now_ms();
            switch (sp.kind) {
                case InitKind::WEIGHT: fill_normal(tmp.data(), n, s, 0.f, 1.0f / sqrtf((float)sp.fan_in)); break;
                case InitKind::BIAS: fill_normal(tmp.data(), n, s, 0.f, 0.02f); break;
                case InitKind::NORM_SCALE: fill_normal(tmp.data(), n, s, 1.f, 0.05f); break;
                case InitKind::ZERO: std::fill(tmp.begin(), tmp.begin() + n, 0.f); break;
            }
            const double t1 = now_ms();
            upload_f32(sp.tensor, tmp.data(), conv);
            t_fill += t1 - t0;
            t_up += now_ms() - t1;
        }
        if (getenv("SDCPP_INIT_TIMING")) fprintf(stderr, "[sd-mi355x] synthetic init: %zu tensors, fill %.0f ms, convert + upload %.0f ms\n", ps.specs.size(), t_fill, t_up);
        return true;
    }

    void upload_f32(ggml_tensor* t, const float* src, std::vector<uint8_t>& scratch) {
        const int64_t n = ggml_nelements(t);
        if (t->type == GGML_TYPE_F32) {
            ggml_backend_tensor_set(t, src, 0, n * 4);
            return;
        }
        scratch.resize(ggml_nbytes(t));
        // model_loader.cpp:168-202; rows are independent, so big tensors convert on a few threads (the software f32 -> f16 / q8_0 / q4_0
        // row encoders run at a few ns per element: 4 s of a 7 s SD1.5 context creation, minutes for a 12 B-parameter FLUX)
        const int64_t rows = n / t->ne[0], per_row = t->ne[0];
        const ggml_type ty = t->type;
        uint8_t* dst       = scratch.data();
        if (ty == GGML_TYPE_F16 || ty == GGML_TYPE_BF16) {
            // elementwise encodings: convert flat ranges (conv kernels have ne0 = 3, a row-wise walk would pay one call per 3 elements)
            parallel_chunks((size_t)n, [&](size_t i0, size_t i1) {
                if (ty == GGML_TYPE_F16)
                    ggml_fp32_to_fp16_row(src + i0, (ggml_fp16_t*)dst + i0, (int64_t)(i1 - i0));
                else
                    ggml_fp32_to_bf16_row(src + i0, (ggml_bf16_t*)dst + i0, (int64_t)(i1 - i0));
            }, 1 << 20);
        } else if (n < (1 << 20)) {
            ggml_quantize_chunk(ty, src, dst, 0, rows, per_row, nullptr);
        } else {
            parallel_chunks((size_t)rows, [&](size_t r0, size_t r1) { ggml_quantize_chunk(ty, src, dst, (int64_t)r0 * per_row, (int64_t)(r1 - r0), per_row, nullptr); }, 2);
        }
        ggml_backend_tensor_set(t, scratch.data(), 0, scratch.size());
    }

    // ---- one cached graph (OUR host-side optimisation; the reference rebuilds and re-allocates per call, ggml_extend.hpp:2767-2930):
    // consecutive calls with the same signature (shapes + flags) reuse the built graph and its gallocr placement — only the inputs are
    // uploaded again.  Building the 2570-node SD1.5 UNet graph and placing it costs 1.5–2.3 ms of host time per step, which on the
    // synchronous path sits between two device steps.  Any other graph allocated from this runner's gallocr invalidates the entry.
    std::string cache_sig;
    ggml_context* cache_ctx = nullptr;
    ggml_cgraph* cache_gf   = nullptr;
    ggml_tensor* cache_res  = nullptr;
    std::vector<HostInput> cache_inputs, cache_builtin;
    int64_t cache_hits = 0;
    void drop_cache() {
        if (cache_ctx) ggml_free(cache_ctx);
        cache_ctx = nullptr;
        cache_gf  = nullptr;
        cache_res = nullptr;
        cache_inputs.clear();
        cache_builtin.clear();
        cache_sig.clear();
    }

    // build -> alloc -> upload inputs -> compute -> download   (ggml_extend.hpp:2767-2930)
    // out == nullptr: nothing is downloaded and nothing waits — inputs go through set_tensor_async and the graph through
    // graph_compute_async on the backend's stream, so the host can build the next graph while this one runs (device-resident sampler)
    // sig / ptrs: non-empty sig enables the cached graph; ptrs are this call's host buffers in the order `build` declares its inputs
    template <typename BuildFn>
    bool compute(BuildFn&& build, float* out, size_t out_bytes, const std::string& sig = std::string(), const std::vector<const void*>& ptrs = {}) {
        ggml_context* cctx = nullptr;
        ggml_cgraph* gf    = nullptr;
        ggml_tensor* res   = nullptr;
        std::vector<HostInput> local_inputs, local_builtin;  // builtin: the runner's own two leaves (uploaded with the inputs, not part of the caller's pointer list)
        std::vector<HostInput>* inputs = &local_inputs;
        const bool cacheable = !sig.empty();
        double t_s = now_ms();
        if (cacheable && cache_ctx && cache_sig == sig && cache_inputs.size() == ptrs.size()) {
            cctx = cache_ctx;
            gf   = cache_gf;
            res  = cache_res;
            for (size_t i = 0; i < ptrs.size(); ++i) cache_inputs[i].data = ptrs[i];
            inputs = &cache_inputs;
            ++cache_hits;
        } else {
            drop_cache();  // whatever is placed next reuses the gallocr buffer
            ggml_init_params ip{0, nullptr, true};
            cctx = ggml_init(ip);
            gf   = ggml_new_graph_custom(cctx, graph_size, false);
            GraphCtx g;
            g.ctx            = cctx;
            g.backend        = backend;
            const double t_b = now_ms();
            res              = build(g, local_inputs);
            // GGMLRunner::get_compute_graph (ggml_extend.hpp:2040-2064) only NAMES the last node; it does not flag it GGML_TENSOR_FLAG_OUTPUT (round 6: found
            // by comparing with the graph the reference's own runner emits, tests/test_ref_graphs.py) and appends two built-in one-element leaves
            ggml_set_name(res, "ggml_runner_final_result_tensor");
            ggml_build_forward_expand(gf, res);
            {
                ggml_tensor* one = ggml_new_tensor_1d(cctx, GGML_TYPE_F32, 1);  // prepare_build_in_tensor_before / _after, ggml_extend.hpp:2023-2036
                ggml_set_name(one, "ggml_runner_build_in_tensor:one");
                ggml_set_input(one);  // set_backend_tensor_data marks what it uploads (ggml_extend.hpp:3090-3095)
                ggml_tensor* zero_int = ggml_new_tensor_1d(cctx, GGML_TYPE_I32, 1);
                ggml_set_name(zero_int, "ggml_runner_build_in_tensor:zero_int");
                ggml_set_input(zero_int);
                ggml_build_forward_expand(gf, one);
                ggml_build_forward_expand(gf, zero_int);
                static const float one_v    = 1.f;
                static const int32_t zero_v = 0;
                local_builtin.push_back(HostInput{one, &one_v, sizeof(one_v)});
                local_builtin.push_back(HostInput{zero_int, &zero_v, sizeof(zero_v)});
            }
            const double t_a = now_ms();
            build_ms += t_a - t_b;
            if (!galloc) galloc = ggml_gallocr_new(ggml_backend_get_default_buffer_type(backend));
            const bool ok = ggml_gallocr_alloc_graph(galloc, gf);
            t_s           = now_ms();
            alloc_ms += t_s - t_a;
            if (!ok) {
                set_error("compute buffer allocation failed");
                ggml_free(cctx);
                return false;
            }
            if (cacheable) {
                bool same = ptrs.size() == local_inputs.size();
                for (size_t i = 0; same && i < ptrs.size(); ++i) same = ptrs[i] == local_inputs[i].data;
                if (same) {  // the caller's pointer list matches what the builder declared: safe to replay with new pointers
                    cache_sig    = sig;
                    cache_ctx    = cctx;
                    cache_gf     = gf;
                    cache_res    = res;
                    cache_inputs = local_inputs;
                    inputs       = &cache_inputs;
                    cache_builtin = local_builtin;
                }
            }
        }
        const bool keep  = cctx == cache_ctx;
        const bool async = out == nullptr;
        if (g_graph_capture) g_last_graph = describe_graph(gf);
        if (g_graph_capture == 2) {
            set_error("graph captured, compute skipped (sdm_set_graph_capture(2))");
            if (keep)
                drop_cache();
            else
                ggml_free(cctx);
            return false;
        }
        for (const std::vector<HostInput>* list : {(const std::vector<HostInput>*)inputs, (const std::vector<HostInput>*)(keep ? &cache_builtin : &local_builtin)})
            for (auto& in : *list) {
                if (!in.t->data && !in.t->buffer) continue;  // a leaf nothing reads is not placed
                if (async)
                    ggml_backend_tensor_set_async(backend, in.t, in.data, 0, in.nbytes);
                else
                    ggml_backend_tensor_set(in.t, in.data, 0, in.nbytes);
            }
        // GGMLRunner::compute routes through the eval-callback variant whenever a callback is installed (ggml_extend.hpp:2857-2860)
        const enum ggml_status st = g_eval_cb ? (enum ggml_status)sdm_backend_graph_compute_with_eval_callback(backend, gf, g_eval_cb, g_eval_cb_data)
                                    : async   ? ggml_backend_graph_compute_async(backend, gf)
                                              : ggml_backend_graph_compute(backend, gf);
        if (st != GGML_STATUS_SUCCESS) {
            set_error(std::string("graph compute failed: ") + ggml_status_to_string(st));
            if (keep)
                drop_cache();
            else
                ggml_free(cctx);
            return false;
        }
        submit_ms += now_ms() - t_s;  // synchronous path: includes the device time of the graph
        if (!async) {
            if (ggml_nbytes(res) != out_bytes) {
                set_error("output size mismatch");
                if (keep)
                    drop_cache();
                else
                    ggml_free(cctx);
                return false;
            }
            ggml_backend_tensor_get(res, out, 0, out_bytes);
        }
        last_nodes = gf->n_nodes;
        ++calls;
        if (!keep) ggml_free(cctx);
        return true;
    }
};

struct sdm_ctx_t {
    sdm_ctx_params_t params;
    ggml_backend_t backend = nullptr;
    Runner unet_runner, vae_runner;
    UNetModel unet;
    MMDiTModel mmdit;
    FluxModel flux;
    bool is_dit = false, is_flux = false;
    float guidance = 3.5f;
    float vae_conv2d_scale = 1.f;  // AutoEncoderKL::set_conv2d_scale (1/32 for SDXL, as the reference sets it without an external VAE)
    FluxFlowDenoiser flux_denoiser;
    VaeDecoder vae;
    CompVisDenoiser denoiser;
    DiscreteFlowDenoiser flow_denoiser;
    int in_channels() const { return is_flux ? 16 : (is_dit ? (int)mmdit.cfg.in_channels : unet.cfg.in_channels); }
    int out_channels() const { return is_flux ? 16 : (is_dit ? (int)mmdit.cfg.out_channels : unet.cfg.out_channels); }
    // Denoiser::get_sigmas (denoiser.hpp:1046-1123): the scheduler sees this family's sigma_min / sigma_max / t_to_sigma.  sched = SCHED_COUNT: the family's own ladder
    std::vector<float> get_sigmas(uint32_t n, int image_seq_len, int sched = SCHED_COUNT) const {
        if (sched == SCHED_COUNT) sched = is_flux ? SCHED_FLUX : SCHED_DISCRETE;
        if (sched == SCHED_FLUX) return flux_denoiser.get_sigmas(n, image_seq_len);  // FluxScheduler needs only the sequence length
        if (is_flux) return scheduler_sigmas(sched, n, flux_denoiser.sigma_min(), flux_denoiser.sigma_max(), [&](float t) { return flux_denoiser.t_to_sigma(t); }, -1);
        if (is_dit) return scheduler_sigmas(sched, n, flow_denoiser.sigma_min(), flow_denoiser.sigma_max(), [&](float t) { return flow_denoiser.t_to_sigma(t); }, -1);
        const int ays = (params.model == SD_MODEL_SD15 || params.model == SD_MODEL_SD15_TINY) ? 0 : 1;  // AYSScheduler picks its table by version (denoiser.hpp:186-200)
        return scheduler_sigmas(sched, n, denoiser.sigma_min(), denoiser.sigma_max(), [&](float t) { return denoiser.t_to_sigma(t); }, ays);
    }
    void scalings(float sigma, float& c_skip, float& c_out, float& c_in) const {
        if (is_flux)
            flux_denoiser.scalings(sigma, c_skip, c_out, c_in);
        else if (is_dit)
            flow_denoiser.scalings(sigma, c_skip, c_out, c_in);
        else if (prediction == SDM_V_PRED) {  // CompVisVDenoiser::get_scalings (denoiser.hpp:1198-1205), sigma_data = 1
            const float sigma_data = 1.0f;
            c_skip = sigma_data * sigma_data / (sigma * sigma + sigma_data * sigma_data);
            c_out  = -sigma * sigma_data / std::sqrt(sigma * sigma + sigma_data * sigma_data);
            c_in   = 1.0f / std::sqrt(sigma * sigma + sigma_data * sigma_data);
        } else
            denoiser.scalings(sigma, c_skip, c_out, c_in);
    }
    int prediction = SDM_EPS_PRED;  // sd_ctx_params_t::prediction (stable-diffusion.h): the UNet families' parameterisation
    float sigma_to_t(float sigma) const { return is_flux ? sigma : (is_dit ? flow_denoiser.sigma_to_t(sigma) : denoiser.sigma_to_t(sigma)); }
    sd_stats_t stats{};
    std::vector<std::pair<std::string, ggml_tensor*>> all_tensors;
    // text encoders: built on first use (sd_text_encoders_init / sd_get_learned_condition / a checkpoint naming them)
    struct TextEncoders {
        Runner l_runner, g_runner, t5_runner;
        ClipTextModel clip_l, clip_g;
        T5Model t5;
        ConditionerSpec spec;
        Runner* runner(int which) { return which == 0 ? &l_runner : (which == 1 ? &g_runner : &t5_runner); }
    };
    std::unique_ptr<TextEncoders> te;
    // device-resident sampler state: the latent batch and the current step's noise live in one persistent buffer
    struct SamplerState {
        ggml_context* sctx        = nullptr;
        ggml_backend_buffer_t buf = nullptr;
        ggml_tensor *x = nullptr, *noise = nullptr, *eps = nullptr;
        int64_t W = 0, H = 0, C = 0, N = 0;
        ~SamplerState() {
            if (buf) ggml_backend_buffer_free(buf);
            if (sctx) ggml_free(sctx);
        }
    };
    std::unique_ptr<SamplerState> sstate;
    // CFG-pair split (sd_set_pair_exchange): this context computes one branch; the per-step sum with the partner is the caller's collective
    sd_pair_exchange_fn pair_fn = nullptr;
    void* pair_user             = nullptr;
    int pair_branch             = 0;
    Runner pair_runner;  // the sampler-update graph behind the exchange keeps its own cached graph + compute buffer
    std::vector<float> pe_cache;  // FLUX rotary table of the last (h, w, n_tokens)
    int pe_h = 0, pe_w = 0;
    int64_t pe_tokens = 0;
    // KL-VAE encoder (auto_encoder_kl.hpp:276-366), made on first sd_vae_encode: the reference builds it unless vae_decode_only is set (stable-diffusion.cpp:1467-1474)
    Runner vae_enc_runner;
    VaeEncoder vae_enc;
    bool vae_enc_ready = false;
    // TAESD (tae.hpp): the tiny decoder next to the KL-VAE, made on first use (sd_tae_decode / sd_use_tae) like the reference makes it when a taesd file is given
    Runner tae_runner;
    TaeDecoder tae;
    bool tae_ready = false, tae_for_images = false;
    std::vector<Runner*> runners() {
        std::vector<Runner*> v{&unet_runner, &vae_runner};
        if (tae_ready) v.push_back(&tae_runner);
        if (vae_enc_ready) v.push_back(&vae_enc_runner);
        if (te) {
            if (te->spec.has_l) v.push_back(&te->l_runner);
            if (te->spec.has_g) v.push_back(&te->g_runner);
            if (te->spec.has_t5) v.push_back(&te->t5_runner);
        }
        return v;
    }
    ~sdm_ctx_t() {
        // runners free their buffers in their destructors; the backend must outlive them
    }
};

// ---------------------------------------------------------------------------------------------------
extern "C" {

const char* sd_last_error(void) { return g_last_error.c_str(); }

bool sd_load_backend(const char* path) { return ggml_backend_load(path) != nullptr; }
int sd_device_count(void) { return (int)ggml_backend_dev_count(); }
const char* sd_device_name(int i) {
    ggml_backend_dev_t d = ggml_backend_dev_get(i);
    return d ? ggml_backend_dev_name(d) : nullptr;
}
const char* sd_device_description(int i) {
    ggml_backend_dev_t d = ggml_backend_dev_get(i);
    return d ? ggml_backend_dev_description(d) : nullptr;
}

void sdm_ctx_params_init(sdm_ctx_params_t* p) {
    memset(p, 0, sizeof(*p));
    p->backend              = nullptr;
    p->model                = SD_MODEL_SD15;
    p->wtype                = SDM_TYPE_F16;
    p->diffusion_flash_attn = false;
    p->vae_decode_only      = true;
    p->weight_seed          = 1234;
    p->n_threads            = -1;
}
void sdm_sample_params_init(sdm_sample_params_t* p) {  // stable-diffusion.cpp:3650-3667
    memset(p, 0, sizeof(*p));
    p->txt_cfg       = 7.0f;
    p->scheduler     = SDM_SCHEDULER_COUNT;  // resolved per model and method at sampling time (sd_get_default_scheduler)
    p->sample_method = SDM_SAMPLE_METHOD_COUNT;  // resolved per family at sampling time (sd_get_default_sample_method)
    p->sample_steps  = 20;
    p->eta           = INFINITY;
    p->flow_shift    = INFINITY;  // the family's default (stable-diffusion.cpp:3665, 3106-3115)
    p->apg_eta         = 1.0f;    // AdaptiveProjectedGuidanceParams defaults (guidance.h:21-26): all neutral = plain classifier-free guidance
    p->slg_layer_start = 0.01f;   // stable-diffusion.cpp:3655-3658
    p->slg_layer_end   = 0.2f;
}
void sdm_img_gen_params_init(sdm_img_gen_params_t* p) {  // stable-diffusion.cpp:3710-3731
    memset(p, 0, sizeof(*p));
    sdm_sample_params_init(&p->sample_params);
    p->width       = 512;
    p->height      = 512;
    p->seed        = 42;
    p->batch_count = 1;
    p->decode      = true;
    p->strength    = 0.75f;  // stable-diffusion.cpp:3718 (read only with an init latent)
}

// locate libggml-mi355x.so next to this library (the GGML_BACKEND_DL convention:
// "a shared object libggml-<name>.so next to the binary", SURVEY.md §8(b) Discovery)
static std::string self_dir() {
    Dl_info info;
    if (dladdr((void*)&self_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s      = p.find_last_of('/');
        return s == std::string::npos ? "." : p.substr(0, s);
    }
    return ".";
}

sdm_ctx_t* sdm_new_ctx(const sdm_ctx_params_t* params) {
    const char* dev_name = params->backend ? params->backend : "MI355X0";
    ggml_backend_dev_t dev = ggml_backend_dev_by_name(dev_name);
    if (!dev && !params->backend) {
        // default device: load the product backend plug-in; no CPU fallback exists in the product
        const std::string path = self_dir() + "/libggml-mi355x.so";
        if (!ggml_backend_load(path.c_str())) {
            set_error("MI355X backend plug-in missing or failed to load: " + path);
            return nullptr;
        }
        dev = ggml_backend_dev_by_name(dev_name);
    }
    if (!dev) {
        set_error(std::string("no ggml device named '") + dev_name + "' (is a gfx950 GPU visible?)");
        return nullptr;
    }
    ggml_backend_t backend = ggml_backend_dev_init(dev, nullptr);
    if (!backend) {
        set_error("ggml_backend_dev_init failed");
        return nullptr;
    }
    sdm_ctx_t* ctx = new sdm_ctx_t();
    ctx->params   = *params;
    ctx->backend  = backend;

    const bool xl   = params->model == SD_MODEL_SDXL || params->model == SD_MODEL_SDXL_TINY;
    const bool tiny = params->model == SD_MODEL_SD15_TINY || params->model == SD_MODEL_SDXL_TINY;
    const bool flux     = params->model == SD_MODEL_FLUX_DEV || params->model == SD_MODEL_FLUX_TINY || params->model == SD_MODEL_FLUX_WIDE1 || params->model == SD_MODEL_FLUX_WIDE8;
    const bool dit      = params->model == SD_MODEL_SD35_LARGE || params->model == SD_MODEL_SD35_TINY || params->model == SD_MODEL_SD35_WIDE2 || params->model == SD_MODEL_SD35_WIDE8 || params->model == SD_MODEL_SD3M_TINY || flux;
    const bool dit_tiny = params->model == SD_MODEL_SD35_TINY || params->model == SD_MODEL_FLUX_TINY || params->model == SD_MODEL_SD3M_TINY;
    UNetConfig ucfg = tiny ? UNetConfig::tiny(xl) : (xl ? UNetConfig::sdxl_base() : UNetConfig::sd15());
    VaeConfig vcfg  = (tiny || dit_tiny) ? VaeConfig::tiny() : (xl ? VaeConfig::sdxl() : VaeConfig::sd15());
    if (tiny && xl) vcfg.scale_factor = 0.13025f;
    if (dit) {  // SD3 VAE: 16 latent channels, no post_quant_conv (auto_encoder_kl.hpp:548-556, 682-684)
        vcfg.z_channels   = 16;
        vcfg.use_quant    = false;
        vcfg.scale_factor = flux ? 0.3611f : 1.5305f;  // auto_encoder_kl.hpp:682-687
        vcfg.shift_factor = flux ? 0.1159f : 0.0609f;
    }

    ctx->unet_runner.backend        = backend;
    ctx->unet_runner.ps.linear_type = (ggml_type)params->wtype;
    ctx->is_dit                     = dit;
    ctx->is_flux                    = flux;
    if (flux) {
        ctx->unet_runner.graph_size = 32768 * 4;  // FLUX_GRAPH_SIZE headroom
        ctx->flux.init(ctx->unet_runner.ps, "model.diffusion_model.", dit_tiny ? FluxConfig::tiny() : (params->model == SD_MODEL_FLUX_WIDE1 ? FluxConfig::flux_wide1() : (params->model == SD_MODEL_FLUX_WIDE8 ? FluxConfig::flux_wide8() : FluxConfig::flux_dev())));
    } else if (dit) {
        ctx->unet_runner.graph_size = 10240 * 8;  // MMDIT_GRAPH_SIZE (mmdit.hpp:14) x our batch headroom
        ctx->mmdit.init(ctx->unet_runner.ps, "model.diffusion_model.", dit_tiny ? (params->model == SD_MODEL_SD3M_TINY ? MMDiTConfig::tiny_medium() : MMDiTConfig::tiny()) : (params->model == SD_MODEL_SD35_WIDE2 ? MMDiTConfig::sd35_wide2() : (params->model == SD_MODEL_SD35_WIDE8 ? MMDiTConfig::sd35_wide8() : MMDiTConfig::sd35_large())));
    } else
        ctx->unet.init(ctx->unet_runner.ps, "model.diffusion_model.", ucfg);  // prefix: stable-diffusion.cpp:1337
    ctx->pair_runner.backend       = backend;
    ctx->pair_runner.graph_size    = 256;
    ctx->vae_runner.backend        = backend;
    ctx->vae_runner.ps.linear_type = GGML_TYPE_F16;
    ctx->vae_runner.graph_size     = 20480;
    ctx->vae.init(ctx->vae_runner.ps, "first_stage_model.", vcfg);  // prefix: stable-diffusion.cpp:1472
    if (xl) {  // SDXL without an external VAE: Conv2d scale 1/32 on every VAE conv (src/stable-diffusion.cpp:1477-1485)
        ctx->vae_conv2d_scale = 1.f / 32.f;
        ctx->vae.set_conv2d_scale(ctx->vae_conv2d_scale);
    }

    if (!ctx->unet_runner.alloc_weights(params->weight_seed) || !ctx->vae_runner.alloc_weights(params->weight_seed)) {
        set_error("weight buffer allocation failed");
        sdm_free_ctx(ctx);
        return nullptr;
    }
    for (auto* r : {&ctx->unet_runner, &ctx->vae_runner})
        for (auto& sp : r->ps.specs) ctx->all_tensors.push_back({sp.name, sp.tensor});
    ctx->stats.weight_bytes = ggml_backend_buffer_get_size(ctx->unet_runner.weights) + ggml_backend_buffer_get_size(ctx->vae_runner.weights);
    return ctx;
}

void sdm_free_ctx(sdm_ctx_t* ctx) {
    if (!ctx) return;
    ggml_backend_t b = ctx->backend;
    delete ctx;
    ggml_backend_free(b);
}

int64_t sd_tensor_count(sdm_ctx_t* ctx) { return (int64_t)ctx->all_tensors.size(); }
const char* sd_tensor_name(sdm_ctx_t* ctx, int64_t i) { return ctx->all_tensors[i].first.c_str(); }
static ggml_tensor* find_tensor(sdm_ctx_t* ctx, const char* name) {
    for (Runner* r : ctx->runners()) {
        auto it = r->ps.by_name.find(name);
        if (it != r->ps.by_name.end()) return it->second;
    }
    return nullptr;
}
bool sd_tensor_info(sdm_ctx_t* ctx, const char* name, int64_t* ne, int* type, size_t* nbytes) {
    ggml_tensor* t = find_tensor(ctx, name);
    if (!t) return false;
    for (int i = 0; i < 4; ++i) ne[i] = t->ne[i];
    *type   = (int)t->type;
    *nbytes = ggml_nbytes(t);
    return true;
}
bool sd_get_tensor(sdm_ctx_t* ctx, const char* name, void* dst, size_t nbytes) {
    ggml_tensor* t = find_tensor(ctx, name);
    if (!t || nbytes != ggml_nbytes(t)) return false;
    ggml_backend_tensor_get(t, dst, 0, nbytes);
    return true;
}
bool sd_get_tensor_f32(sdm_ctx_t* ctx, const char* name, float* dst, int64_t nelem) {
    ggml_tensor* t = find_tensor(ctx, name);
    if (!t || nelem != ggml_nelements(t)) return false;
    std::vector<uint8_t> raw(ggml_nbytes(t));
    ggml_backend_tensor_get(t, raw.data(), 0, raw.size());
    const int64_t rows = nelem / t->ne[0];
    const size_t rs    = ggml_row_size(t->type, t->ne[0]);
    for (int64_t r = 0; r < rows; ++r) ggml_dequantize_row(t->type, raw.data() + r * rs, dst + r * t->ne[0], t->ne[0]);
    return true;
}
bool sd_set_tensor_f32(sdm_ctx_t* ctx, const char* name, const float* src, int64_t nelem) {
    ggml_tensor* t = find_tensor(ctx, name);
    if (!t || nelem != ggml_nelements(t)) return false;
    std::vector<uint8_t> scratch;
    ctx->unet_runner.upload_f32(t, src, scratch);
    return true;
}

// ---- checkpoint files -> weight buffers ---------------------------------------------
// ModelLoader::load_tensors (src/model_loader.cpp:1180-1260): for every tensor the model declares that the file holds, convert
// file dtype -> f32 -> the parameter's ggml type (convert_tensor, model_loader.cpp:155-205) and ggml_backend_tensor_set it.
// Returns the number of parameters loaded, or -1 on a file / shape error; tensors the file does not name keep their current values.
static bool ensure_text_encoders(sdm_ctx_t* ctx);
static bool ensure_vae_encoder(sdm_ctx_t* ctx);
static bool ensure_tae(sdm_ctx_t* ctx);

static NameDialect name_dialect(const sdm_ctx_t* ctx) {
    NameDialect d;
    d.unet_family = !ctx->is_dit;
    d.flux        = ctx->is_flux;
    if (!ctx->is_dit) {
        const UNetConfig& c = ctx->unet.cfg;
        d.unet_levels       = (int)c.channel_mult.size();
        d.unet_res_blocks   = c.num_res_blocks;
        d.unet_attn_levels.clear();
        int ds = 1;
        for (int l = 0; l < d.unet_levels; ++l, ds *= 2)
            if (std::find(c.attention_resolutions.begin(), c.attention_resolutions.end(), ds) != c.attention_resolutions.end()) d.unet_attn_levels.push_back(l);
    }
    d.vae_levels = (int)ctx->vae.cfg.ch_mult.size();
    return d;
}

bool sd_convert_tensor_name(sdm_ctx_t* ctx, const char* name, char* out, size_t out_capacity) {
    const std::string r = canonical_tensor_name(name, name_dialect(ctx));
    if (r.size() + 1 > out_capacity) return false;
    memcpy(out, r.c_str(), r.size() + 1);
    return true;
}

int64_t sd_load_weights(sdm_ctx_t* ctx, const char* path, int64_t* n_missing, int64_t* n_unused) { return sd_load_weights_prefixed(ctx, path, nullptr, n_missing, n_unused); }

int64_t sd_load_weights_prefixed(sdm_ctx_t* ctx, const char* path, const char* prefix, int64_t* n_missing, int64_t* n_unused) {
    ModelFile mf;
    if (!read_model_file(path, mf)) {
        set_error(mf.error);
        return -1;
    }
    // file names -> canonical names (init_from_file_and_convert_name, model_loader.cpp); OpenCLIP's fused in_proj rows are split into the
    // q / k / v projections the graph registers: rows [0,E) [E,2E) [2E,3E) of the [3E, E] weight (the reference keeps it fused and
    // chunks the activation instead, ggml_extend.hpp:4074-4081 — the same dot products)
    const NameDialect dialect = name_dialect(ctx);
    std::vector<FileTensor> expanded;
    expanded.reserve(mf.tensors.size());
    bool names_te = false, names_enc = false, names_tae = false;
    for (const FileTensor& t : mf.tensors) {
        FileTensor c = t;
        c.name       = canonical_tensor_name(prefix ? std::string(prefix) + t.name : t.name, dialect);
        names_te     = names_te || c.name.rfind("cond_stage_model.", 0) == 0 || c.name.rfind("text_encoders.", 0) == 0;
        names_enc    = names_enc || c.name.rfind("first_stage_model.encoder.", 0) == 0;
        names_tae    = names_tae || c.name.rfind("tae.decoder.layers.", 0) == 0;
        const std::vector<std::string> qkv = split_in_proj_names(c.name);
        const int outer                    = c.n_dims >= 2 ? 1 : 0;  // the fused dimension: rows of the weight, elements of the bias
        if (!qkv.empty() && c.ne[outer] % 3 == 0 && (outer == 1 || !ggml_is_quantized(c.type))) {
            for (int k = 0; k < 3; ++k) {
                FileTensor part = c;
                part.name       = qkv[k];
                part.ne[outer]  = c.ne[outer] / 3;
                part.nbytes     = c.nbytes / 3;
                part.offset     = c.offset + (uint64_t)k * part.nbytes;
                expanded.push_back(part);
            }
        } else {
            expanded.push_back(c);
        }
    }
    std::map<std::string, const FileTensor*> dir;
    for (auto& t : expanded) dir[t.name] = &t;
    std::map<std::string, std::string> undecodable;  // canonical name -> dtype the readers could not decode
    for (auto& kv : mf.undecodable) undecodable[canonical_tensor_name(prefix ? std::string(prefix) + kv.first : kv.first, dialect)] = kv.second;
    if (names_te && !ensure_text_encoders(ctx)) return -1;  // like the conditioners' tensor_storage_map probes (conditioner.hpp:630-651)
    // a checkpoint that carries the VAE encoder / a taesd file: the modules made on first use are made now, so their tensors are loaded instead of reported unused
    if (names_enc && !ensure_vae_encoder(ctx)) return -1;
    if (names_tae && !ensure_tae(ctx)) return -1;
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_error(std::string("cannot open ") + path);
        return -1;
    }
    int64_t loaded = 0, missing = 0;
    std::vector<uint8_t> raw, conv;
    std::vector<float> f32;
    std::map<std::string, bool> used;
    for (auto& nt : ctx->all_tensors) {
        auto it = dir.find(nt.first);
        if (it == dir.end()) {
            auto ud = undecodable.find(nt.first);
            if (ud != undecodable.end()) {  // the file HAS this parameter, in a dtype we cannot read: leaving the synthetic weights in place would be silent garbage
                set_error("tensor '" + nt.first + "' is stored as " + ud->second + ", which this build cannot decode");
                fclose(f);
                return -1;
            }
            ++missing;
            continue;
        }
        ggml_tensor* t       = nt.second;
        const int64_t n      = ggml_nelements(t);
        // a fused parameter the file keeps as separate tensors (diffusers DiT q / k / v [/ proj_mlp], name_conversion.hpp: "<name>", "<name>.1",
        // "<name>.2", ...): stack the parts along the output dimension (rows of a weight, elements of a bias)
        if (dir.count(nt.first + ".1")) {
            std::vector<const FileTensor*> parts{it->second};
            for (int k = 1;; ++k) {
                auto pk = dir.find(nt.first + "." + std::to_string(k));
                if (pk == dir.end()) break;
                parts.push_back(pk->second);
            }
            const int64_t inner = ggml_n_dims(t) >= 2 ? t->ne[0] : 1, outer_total = n / inner;
            int64_t outer_sum = 0;
            bool ok = ggml_n_dims(t) <= 2, native_same = true;
            for (const FileTensor* pt : parts) {
                int64_t pn = 1;
                for (int d = 0; d < 4; ++d) pn *= pt->ne[d];
                ok = ok && pn % inner == 0 && (inner == 1 || pt->ne[0] == inner) && pt->n_dims <= 2;
                const uint64_t want = pt->kind != SrcKind::NATIVE ? pt->nbytes : (uint64_t)ggml_row_size(pt->type, pt->ne[0]) * (uint64_t)(pn / pt->ne[0]);
                ok = ok && pt->nbytes == want;
                outer_sum += pn / inner;
                native_same = native_same && pt->kind == SrcKind::NATIVE && pt->type == t->type;
            }
            if (!ok || outer_sum != outer_total) {
                set_error("fused parts do not add up to the shape of " + nt.first);
                fclose(f);
                return -1;
            }
            if (!native_same) f32.assign((size_t)n, 0.f);
            int64_t o0 = 0;  // first output row of the current part
            for (size_t k = 0; k < parts.size(); ++k) {
                const FileTensor& pt = *parts[k];
                int64_t pn = 1;
                for (int d = 0; d < 4; ++d) pn *= pt.ne[d];
                raw.resize(pt.nbytes);
                if (fseek(f, (long)pt.offset, SEEK_SET) != 0 || fread(raw.data(), 1, pt.nbytes, f) != pt.nbytes) {
                    set_error("short read for " + nt.first);
                    fclose(f);
                    return -1;
                }
                if (native_same) {
                    ggml_backend_tensor_set(t, raw.data(), (size_t)o0 * ggml_row_size(t->type, inner), pt.nbytes);
                } else if (pt.kind != SrcKind::NATIVE) {
                    decode_src_kind(pt.kind, raw.data(), pn, f32.data() + o0 * inner);
                } else {
                    const int64_t rows = pn / pt.ne[0];
                    const size_t rs    = ggml_row_size(pt.type, pt.ne[0]);
                    for (int64_t r = 0; r < rows; ++r) ggml_dequantize_row(pt.type, raw.data() + r * rs, f32.data() + o0 * inner + r * pt.ne[0], pt.ne[0]);
                }
                o0 += pn / inner;
                used[nt.first + (k ? "." + std::to_string(k) : std::string())] = true;
            }
            if (!native_same) ctx->unet_runner.upload_f32(t, f32.data(), conv);
            ++loaded;
            continue;
        }
        const FileTensor& ft = *it->second;
        int64_t fn           = 1;
        for (int d = 0; d < 4; ++d) fn *= ft.ne[d];
        // shapes must agree up to trailing 1s; Linear / conv weights additionally dim by dim (torch [out,in,kh,kw] == ggml [kw,kh,in,out]);
        // a [out, in] matrix may fill a 1x1 conv kernel [1,1,in,out] (diffusers stores the VAE attention projections as Linear)
        bool same = fn == n;
        const bool lin_as_conv1x1 = ft.n_dims == 2 && t->ne[0] == 1 && t->ne[1] == 1 && ft.ne[0] == t->ne[2] && ft.ne[1] == t->ne[3];
        for (int d = 0; same && !lin_as_conv1x1 && d < 4; ++d) same = ft.ne[d] == t->ne[d] || (ft.n_dims <= 2 && ggml_n_dims(t) <= 2);
        if (same && !lin_as_conv1x1 && ft.n_dims <= 2 && ggml_n_dims(t) <= 2) same = ft.ne[0] == t->ne[0];
        if (!same) {
            set_error("shape mismatch for " + nt.first);
            fclose(f);
            return -1;
        }
        // the readers validated nbytes against the file's own shape; here against what the copies below will touch
        const uint64_t want = ft.kind != SrcKind::NATIVE ? ft.nbytes : (uint64_t)ggml_row_size(ft.type, ft.ne[0]) * (uint64_t)(n / ft.ne[0]);
        if (ft.nbytes != want || (ft.type == t->type && ft.kind == SrcKind::NATIVE && ft.nbytes != ggml_nbytes(t))) {
            set_error("size mismatch for tensor '" + nt.first + "'");
            fclose(f);
            return -1;
        }
        raw.resize(ft.nbytes);
        if (fseek(f, (long)ft.offset, SEEK_SET) != 0 || fread(raw.data(), 1, ft.nbytes, f) != ft.nbytes) {
            set_error("short read for " + nt.first);
            fclose(f);
            return -1;
        }
        if (ft.kind != SrcKind::NATIVE) {
            f32.resize(n);
            decode_src_kind(ft.kind, raw.data(), n, f32.data());
            ctx->unet_runner.upload_f32(t, f32.data(), conv);
        } else if (ft.type == t->type) {
            ggml_backend_tensor_set(t, raw.data(), 0, ggml_nbytes(t));
        } else {
            f32.resize(n);
            const int64_t rows = n / ft.ne[0];
            const size_t rs    = ggml_row_size(ft.type, ft.ne[0]);
            for (int64_t r = 0; r < rows; ++r) ggml_dequantize_row(ft.type, raw.data() + r * rs, f32.data() + r * ft.ne[0], ft.ne[0]);
            ctx->unet_runner.upload_f32(t, f32.data(), conv);
        }
        used[nt.first] = true;
        ++loaded;
    }
    fclose(f);
    if (n_missing) *n_missing = missing;
    if (n_unused) *n_unused = (int64_t)expanded.size() - (int64_t)used.size();
    // The reference runs SDXL's VAE with Conv2d scale 1/32 only when NO valid external VAE is given (or --force-sdxl-vae-conv-scale is set): a fixed fp16 VAE
    // loaded through --vae keeps scale 1 (src/stable-diffusion.cpp:1477-1485).  Loading a file under the VAE prefix IS the --vae path here, so a
    // successful load of VAE tensors restores scale 1 (round-5 advice); sd_set_vae_conv2d_scale(ctx, 1/32) afterwards is the force flag.
    if (loaded > 0 && prefix && strncmp(prefix, "first_stage_model", 17) == 0 && ctx->vae_conv2d_scale != 1.f) {
        fprintf(stderr, "[sd-mi355x] external VAE loaded: Conv2d scale %.5f -> 1 (call sd_set_vae_conv2d_scale to force a scale)\n", (double)ctx->vae_conv2d_scale);
        ctx->vae_conv2d_scale = 1.f;
        ctx->vae.set_conv2d_scale(1.f);
        if (ctx->vae_enc_ready) ctx->vae_enc.set_conv2d_scale(1.f);
    }
    return loaded;
}

// ---- text encoders + conditioner ----------------------------------------------------
// Which encoders a family owns, their parameter prefixes and output wiring: FrozenCLIPEmbedderWithCustomWords (conditioner.hpp:169-190),
// SD3CLIPEmbedder (:623-652), FluxCLIPEmbedder (:1034-1062).  Weights are synthetic (weight_seed) until sd_load_weights overwrites them.
static bool ensure_text_encoders(sdm_ctx_t* ctx) {
    if (ctx->te) return true;
    std::unique_ptr<sdm_ctx_t::TextEncoders> te(new sdm_ctx_t::TextEncoders());
    const int m     = (int)ctx->params.model;
    const bool tiny = m == SD_MODEL_SD15_TINY || m == SD_MODEL_SDXL_TINY || m == SD_MODEL_SD35_TINY || m == SD_MODEL_FLUX_TINY || m == SD_MODEL_SD3M_TINY;
    ConditionerSpec& sp = te->spec;
    ClipTextConfig lc, gc;
    T5Config tc;
    std::string lp, gp, tp;
    if (m == SD_MODEL_SD15 || m == SD_MODEL_SD15_TINY) {
        sp.family = CondFamily::SD1;
        lc        = tiny ? ClipTextConfig::tiny(64, 4, 0, false, true) : ClipTextConfig::vit_l(true);
        lp        = "cond_stage_model.transformer.text_model.";
    } else if (m == SD_MODEL_SDXL || m == SD_MODEL_SDXL_TINY) {
        sp.family  = CondFamily::SDXL;
        sp.has_g   = true;
        lc         = tiny ? ClipTextConfig::tiny(24, 2, 0, false, false) : ClipTextConfig::vit_l(false);
        gc         = tiny ? ClipTextConfig::tiny(40, 2, 48, true, false) : ClipTextConfig::vit_bigg(false);
        lp         = "cond_stage_model.transformer.text_model.";
        gp         = "cond_stage_model.1.transformer.text_model.";
        sp.adm_dim = ctx->unet.cfg.adm_in_channels;
        sp.ts_dim  = tiny ? 8 : 256;
    } else if (m == SD_MODEL_SD35_LARGE || m == SD_MODEL_SD35_TINY || m == SD_MODEL_SD35_WIDE2 || m == SD_MODEL_SD35_WIDE8 || m == SD_MODEL_SD3M_TINY) {
        sp.family = CondFamily::SD3;
        sp.has_g = sp.has_t5 = true;
        lc = tiny ? ClipTextConfig::tiny(24, 2, 0, false, false) : ClipTextConfig::vit_l(false);
        gc = tiny ? ClipTextConfig::tiny(40, 2, 40, true, false) : ClipTextConfig::vit_bigg(false);
        tc = tiny ? T5Config::tiny(96) : T5Config::xxl();
        lp = "text_encoders.clip_l.transformer.text_model.";
        gp = "text_encoders.clip_g.transformer.text_model.";
        tp = "text_encoders.t5xxl.transformer.";
    } else {
        sp.family   = CondFamily::FLUX;
        sp.has_t5   = true;
        sp.t5_chunk = tiny ? 32 : 256;
        lc          = tiny ? ClipTextConfig::tiny(64, 4, 0, false, true) : ClipTextConfig::vit_l(true);
        tc          = tiny ? T5Config::tiny(96) : T5Config::xxl();
        lp          = "text_encoders.clip_l.transformer.text_model.";
        tp          = "text_encoders.t5xxl.transformer.";
    }
    sp.l_dim  = lc.hidden_size;
    sp.g_dim  = gc.hidden_size;
    sp.g_proj = gc.projection_dim;
    sp.t5_dim = tc.model_dim;
    sp.eos_id = (int32_t)lc.vocab_size - 1;  // 49407 = <|endoftext|> of the 49408-entry CLIP vocabulary
    auto setup = [&](Runner& r, size_t graph_size) {
        r.backend        = ctx->backend;
        r.ps.linear_type = (ggml_type)ctx->params.wtype;
        r.graph_size     = graph_size;
    };
    setup(te->l_runner, 2048);  // clip.hpp:522
    te->clip_l.init(te->l_runner.ps, lp, lc);
    if (sp.has_g) {
        setup(te->g_runner, 4096);
        te->clip_g.init(te->g_runner.ps, gp, gc);
    }
    if (sp.has_t5) {
        setup(te->t5_runner, 4096);
        te->t5.init(te->t5_runner.ps, tp, tc);
    }
    ctx->te = std::move(te);
    for (Runner* r : ctx->runners()) {
        if (r == &ctx->unet_runner || r == &ctx->vae_runner) continue;
        if (!r->alloc_weights(ctx->params.weight_seed)) {
            set_error("text-encoder weight buffer allocation failed");
            ctx->te.reset();
            return false;
        }
    }
    for (Runner* r : ctx->runners()) {
        if (r == &ctx->unet_runner || r == &ctx->vae_runner) continue;
        for (auto& spn : r->ps.specs) ctx->all_tensors.push_back({spn.name, spn.tensor});
        ctx->stats.weight_bytes += ggml_backend_buffer_get_size(r->weights);
    }
    return true;
}

// CLIPTextModelRunner::build_graph / compute (clip.hpp:516-583)
static bool te_clip_forward(sdm_ctx_t* ctx, int which, const int32_t* ids, int64_t n_tokens, size_t max_token_idx, bool return_pooled, int clip_skip, std::vector<float>& out) {
    if (!ensure_text_encoders(ctx)) return false;
    auto& te = *ctx->te;
    if (which < 0 || which > 1 || (which == 1 && !te.spec.has_g)) {
        set_error("this model family has no such CLIP tower");
        return false;
    }
    const ClipTextModel& m = which == 0 ? te.clip_l : te.clip_g;
    const int64_t L        = m.cfg.n_token;
    if (n_tokens <= 0 || n_tokens % L != 0) {  // clip.hpp:509-512: longer inputs fold into a batch of n_token-long rows
        set_error("CLIP input length must be a positive multiple of 77");
        return false;
    }
    const int64_t N = n_tokens / L;
    for (int64_t i = 0; i < n_tokens; ++i)
        if (ids[i] < 0 || ids[i] >= m.cfg.vocab_size) {
            set_error("token id out of range");
            return false;
        }
    if (max_token_idx >= (size_t)n_tokens) {
        set_error("max_token_idx out of range");
        return false;
    }
    const std::vector<float> mask = ClipTextModel::causal_mask((int)L);
    out.resize((size_t)(return_pooled ? (m.text_projection ? m.cfg.projection_dim : m.cfg.hidden_size) : m.cfg.hidden_size * n_tokens));
    auto build = [&](GraphCtx& g, std::vector<HostInput>& in) {
        g.flash_attn     = ctx->params.diffusion_flash_attn;
        ggml_tensor* tid = ggml_new_tensor_2d(g.ctx, GGML_TYPE_I32, L, N);
        ggml_set_input(tid);
        in.push_back({tid, ids, ggml_nbytes(tid)});
        ggml_tensor* tm = ggml_new_tensor_2d(g.ctx, GGML_TYPE_F32, L, L);
        ggml_set_input(tm);
        in.push_back({tm, mask.data(), ggml_nbytes(tm)});
        return m.forward(g, tid, tm, max_token_idx, return_pooled, clip_skip);
    };
    return te.runner(which)->compute(build, out.data(), out.size() * sizeof(float));
}

// T5Runner::build_graph / compute (t5.hpp:422-461); no padding mask (SD3 / FLUX pass none)
static bool te_t5_forward(sdm_ctx_t* ctx, const int32_t* ids, int64_t n_tokens, std::vector<float>& out) {
    if (!ensure_text_encoders(ctx)) return false;
    auto& te = *ctx->te;
    if (!te.spec.has_t5) {
        set_error("this model family has no T5 encoder");
        return false;
    }
    const T5Model& m = te.t5;
    if (n_tokens <= 0) {
        set_error("empty T5 input");
        return false;
    }
    for (int64_t i = 0; i < n_tokens; ++i)
        if (ids[i] < 0 || ids[i] >= m.cfg.vocab_size) {
            set_error("token id out of range");
            return false;
        }
    const std::vector<int32_t> buckets = T5Model::relative_position_buckets((int)n_tokens, (int)n_tokens);
    out.resize((size_t)(m.cfg.model_dim * n_tokens));
    auto build = [&](GraphCtx& g, std::vector<HostInput>& in) {
        g.flash_attn     = ctx->params.diffusion_flash_attn;
        ggml_tensor* tid = ggml_new_tensor_2d(g.ctx, GGML_TYPE_I32, n_tokens, 1);
        ggml_set_input(tid);
        in.push_back({tid, ids, ggml_nbytes(tid)});
        ggml_tensor* tb = ggml_new_tensor_2d(g.ctx, GGML_TYPE_I32, n_tokens, n_tokens);
        ggml_set_input(tb);
        in.push_back({tb, buckets.data(), ggml_nbytes(tb)});
        return m.forward(g, tid, tb, nullptr);
    };
    return te.t5_runner.compute(build, out.data(), out.size() * sizeof(float));
}

bool sd_text_encoders_init(sdm_ctx_t* ctx) { return ensure_text_encoders(ctx); }

int64_t sd_clip_forward(sdm_ctx_t* ctx, int which, const int32_t* ids, int n_tokens, int max_token_idx, bool return_pooled, int clip_skip, float* out, int64_t out_capacity) {
    std::vector<float> o;
    if (max_token_idx < 0) {
        set_error("max_token_idx out of range");
        return -1;
    }
    if (!te_clip_forward(ctx, which, ids, n_tokens, (size_t)max_token_idx, return_pooled, clip_skip, o)) return -1;
    if ((int64_t)o.size() > out_capacity) {
        set_error("output buffer too small");
        return -1;
    }
    std::copy(o.begin(), o.end(), out);
    return (int64_t)o.size();
}

int64_t sd_t5_forward(sdm_ctx_t* ctx, const int32_t* ids, int n_tokens, float* out, int64_t out_capacity) {
    std::vector<float> o;
    if (!te_t5_forward(ctx, ids, n_tokens, o)) return -1;
    if ((int64_t)o.size() > out_capacity) {
        set_error("output buffer too small");
        return -1;
    }
    std::copy(o.begin(), o.end(), out);
    return (int64_t)o.size();
}

int sd_t5_relative_position_buckets(int q_len, int k_len, int32_t* out) {
    const std::vector<int32_t> b = T5Model::relative_position_buckets(q_len, k_len);
    std::copy(b.begin(), b.end(), out);
    return (int)b.size();
}

bool sd_get_learned_condition(sdm_ctx_t* ctx, const sd_token_list_t* clip_l, const sd_token_list_t* clip_g, const sd_token_list_t* t5, int clip_skip, int width,
                              int height, bool zero_out_masked, float* crossattn_out, int64_t crossattn_capacity, int64_t* crossattn_ne, float* vector_out,
                              int64_t vector_capacity, int64_t* vector_n) {
    if (!ensure_text_encoders(ctx)) return false;
    auto to_list = [](const sd_token_list_t* t) {
        TokenList l;
        if (t && t->n > 0) {
            l.ids.assign(t->ids, t->ids + t->n);
            if (t->weights)
                l.weights.assign(t->weights, t->weights + t->n);
            else
                l.weights.assign((size_t)t->n, 1.0f);
        }
        return l;
    };
    EncoderFns fn;
    fn.clip = [&](int which, const std::vector<int32_t>& ids, size_t max_idx, bool pooled, int skip, std::vector<float>& o) {
        return te_clip_forward(ctx, which, ids.data(), (int64_t)ids.size(), max_idx, pooled, skip, o);
    };
    fn.t5 = [&](const std::vector<int32_t>& ids, std::vector<float>& o) { return te_t5_forward(ctx, ids.data(), (int64_t)ids.size(), o); };
    Condition c;
    std::string err;
    if (!get_learned_condition(ctx->te->spec, fn, to_list(clip_l), to_list(clip_g), to_list(t5), clip_skip, width, height, zero_out_masked, c, err)) {
        if (!err.empty()) set_error(err);
        return false;
    }
    if (crossattn_ne) {
        crossattn_ne[0] = c.ctx_dim;
        crossattn_ne[1] = c.n_tokens;
    }
    if (vector_n) *vector_n = (int64_t)c.vec.size();
    if (crossattn_out) {
        if ((int64_t)c.crossattn.size() > crossattn_capacity) {
            set_error("c_crossattn buffer too small");
            return false;
        }
        std::copy(c.crossattn.begin(), c.crossattn.end(), crossattn_out);
    }
    if (vector_out && !c.vec.empty()) {
        if ((int64_t)c.vec.size() > vector_capacity) {
            set_error("c_vector buffer too small");
            return false;
        }
        std::copy(c.vec.begin(), c.vec.end(), vector_out);
    }
    return true;
}

// ---- one UNet forward ---------------------------------------------------------------------------
// host-built side inputs of one model call (FLUX: guidance vector and rotary table, flux.hpp:1457-1500)
struct ModelSideInputs {
    std::vector<float> guidance;
    const std::vector<float>* pe_table = nullptr;
    const std::vector<float>& pe() const { return *pe_table; }
};
static bool prepare_side_inputs(sdm_ctx_t* ctx, int w, int h, int n, int64_t n_tokens, bool has_y, ModelSideInputs& si) {
    if (!ctx->is_flux) return true;
    if (!has_y) {
        set_error("FLUX needs the pooled text vector y");
        return false;
    }
    si.guidance.assign(n, ctx->guidance);
    // the rotary table depends on the latent size and the text length only: 12 ms of sin/cos per call at 1024x1024 + 256 tokens if rebuilt
    // every forward like the reference does (flux.hpp:1457-1500) — kept per context instead
    if (ctx->pe_cache.empty() || ctx->pe_h != h || ctx->pe_w != w || ctx->pe_tokens != n_tokens) {
        ctx->pe_cache  = gen_flux_pe(h, w, ctx->flux.cfg.patch_size, (int)n_tokens, ctx->flux.cfg.axes_dim, (float)ctx->flux.cfg.theta);
        ctx->pe_h      = h;
        ctx->pe_w      = w;
        ctx->pe_tokens = n_tokens;
    }
    si.pe_table = &ctx->pe_cache;
    return true;
}
// the denoiser network on an existing latent tensor tx [w,h,c,n]: declares the remaining graph inputs and calls the family's forward
static ggml_tensor* build_model_call(sdm_ctx_t* ctx, GraphCtx& g, std::vector<HostInput>& in, ggml_tensor* tx, int n, const float* timesteps, const float* context,
                                     int64_t ctx_dim, int64_t n_tokens, int64_t ctx_n, const float* y, int64_t y_dim, int64_t y_n, const ModelSideInputs& si) {
    g.flash_attn    = ctx->params.diffusion_flash_attn;
    g.conv_direct   = ctx->params.diffusion_conv_direct;
    ggml_tensor* tt = ggml_new_tensor_1d(g.ctx, GGML_TYPE_F32, n);
    ggml_set_input(tt);
    in.push_back({tt, timesteps, ggml_nbytes(tt)});
    ggml_tensor* tc = ggml_new_tensor_3d(g.ctx, GGML_TYPE_F32, ctx_dim, n_tokens, ctx_n);
    ggml_set_input(tc);
    in.push_back({tc, context, ggml_nbytes(tc)});
    ggml_tensor* ty = nullptr;
    if (y != nullptr) {
        ty = ggml_new_tensor_2d(g.ctx, GGML_TYPE_F32, y_dim, y_n);
        ggml_set_input(ty);
        in.push_back({ty, y, ggml_nbytes(ty)});
    }
    if (ctx->is_flux) {
        // guidance [N] and the rotary table are host-built inputs of every call (flux.hpp:1457-1500: pe_vec generated on the CPU and uploaded)
        ggml_tensor* tg = ggml_new_tensor_1d(g.ctx, GGML_TYPE_F32, n);
        ggml_set_input(tg);
        in.push_back({tg, si.guidance.data(), ggml_nbytes(tg)});
        const FluxConfig& fc = ctx->flux.cfg;
        ggml_tensor* tp      = ggml_new_tensor_4d(g.ctx, GGML_TYPE_F32, 2, 2, fc.hidden_size / fc.num_heads / 2, (int64_t)si.pe().size() / (2 * (fc.hidden_size / fc.num_heads)));
        ggml_set_input(tp);
        in.push_back({tp, si.pe().data(), ggml_nbytes(tp)});
        return ctx->flux.forward(g, tx, tt, tc, ty, tg, tp);
    }
    return ctx->is_dit ? ctx->mmdit.forward(g, tx, tt, tc, ty) : ctx->unet.forward(g, tx, tt, tc, ty);
}

// skip: the joint blocks a skip-layer-guidance forward leaves out (MMDiT only; NULL / empty = the whole model)
static bool unet_forward_skip(sdm_ctx_t* ctx, const float* x, int w, int h, int c, int n, const float* timesteps, const float* context, int64_t ctx_dim, int64_t n_tokens,
                              int64_t ctx_n, const float* y, int64_t y_dim, int64_t y_n, float* out, const std::vector<int>* skip) {
    Runner& r = ctx->unet_runner;
    ModelSideInputs si;
    if (!prepare_side_inputs(ctx, w, h, n, n_tokens, y != nullptr, si)) return false;
    if (skip && skip->empty()) skip = nullptr;
    auto build = [&](GraphCtx& g, std::vector<HostInput>& in) {
        ggml_tensor* tx = ggml_new_tensor_4d(g.ctx, GGML_TYPE_F32, w, h, c, n);
        ggml_set_input(tx);
        in.push_back({tx, x, ggml_nbytes(tx)});
        g.skip_layers = skip;
        return build_model_call(ctx, g, in, tx, n, timesteps, context, ctx_dim, n_tokens, ctx_n, y, y_dim, y_n, si);
    };
    char sig[256];
    int so = snprintf(sig, sizeof(sig), "fwd %d %d %d %d %lld %lld %lld %lld %lld %d", w, h, c, n, (long long)ctx_dim, (long long)n_tokens, (long long)ctx_n,
                      (long long)(y ? y_dim : -1), (long long)y_n, (int)ctx->is_flux);
    if (skip)
        for (int l : *skip)
            if (so < (int)sizeof(sig) - 8) so += snprintf(sig + so, sizeof(sig) - so, " s%d", l);
    std::vector<const void*> ptrs{x, timesteps, context};
    if (y) ptrs.push_back(y);
    if (ctx->is_flux) {
        ptrs.push_back(si.guidance.data());
        ptrs.push_back(si.pe().data());
    }
    const bool ok = r.compute(build, out, (size_t)w * h * ctx->out_channels() * n * sizeof(float), sig, ptrs);
    ctx->stats.unet_calls  = r.calls;
    ctx->stats.graph_nodes = r.last_nodes;
    if (r.galloc) ctx->stats.compute_buffer_bytes = ggml_gallocr_get_buffer_size(r.galloc, 0);
    return ok;
}
bool sd_unet_forward(sdm_ctx_t* ctx, const float* x, int w, int h, int c, int n, const float* timesteps, const float* context,
                     int64_t ctx_dim, int64_t n_tokens, int64_t ctx_n, const float* y, int64_t y_dim, int64_t y_n, float* out) {
    return unet_forward_skip(ctx, x, w, h, c, n, timesteps, context, ctx_dim, n_tokens, ctx_n, y, y_dim, y_n, out, nullptr);
}

// the same forward without the listed MMDiT joint blocks (MMDiT::forward's skip_layers, mmdit.hpp:854-866): what skip-layer guidance evaluates
bool sd_unet_forward_skip_layers(sdm_ctx_t* ctx, const float* x, int w, int h, int c, int n, const float* timesteps, const float* context, int64_t ctx_dim, int64_t n_tokens,
                                 int64_t ctx_n, const float* y, int64_t y_dim, int64_t y_n, const int* skip_layers, int n_skip, float* out) {
    if (!ctx->is_dit || ctx->is_flux) {
        set_error("sd_unet_forward_skip_layers: skip layers exist for the MMDiT family only");
        return false;
    }
    const std::vector<int> skip(skip_layers, skip_layers + (n_skip > 0 ? n_skip : 0));
    return unet_forward_skip(ctx, x, w, h, c, n, timesteps, context, ctx_dim, n_tokens, ctx_n, y, y_dim, y_n, out, &skip);
}

// ---- VAE decode ---------------------------------------------------------------------------------
bool sd_vae_decode(sdm_ctx_t* ctx, const float* latents, int w, int h, int c, int n, float* out_rgb) {
    Runner& r       = ctx->vae_runner;
    const float sf  = ctx->vae.cfg.scale_factor, sh = ctx->vae.cfg.shift_factor;
    const size_t ne = (size_t)w * h * c * n;
    std::vector<float> z(ne);
    for (size_t i = 0; i < ne; ++i) z[i] = latents[i] / sf + sh;  // diffusion_to_vae_latents, auto_encoder_kl.hpp:818-826
    auto build = [&](GraphCtx& g, std::vector<HostInput>& in) {
        g.flash_attn    = ctx->params.diffusion_flash_attn;
        g.conv_direct   = ctx->params.diffusion_conv_direct;
        ggml_tensor* tz = ggml_new_tensor_4d(g.ctx, GGML_TYPE_F32, w, h, c, n);
        ggml_set_input(tz);
        in.push_back({tz, z.data(), ggml_nbytes(tz)});
        return ctx->vae.forward(g, tz);
    };
    const size_t on = (size_t)w * 8 * h * 8 * 3 * n;
    const double t0 = now_ms();
    char sig[64];
    snprintf(sig, sizeof(sig), "vae %d %d %d %d s%g", w, h, c, n, (double)ctx->vae_conv2d_scale);
    if (!r.compute(build, out_rgb, on * sizeof(float), sig, {z.data()})) return false;
    parallel_chunks(on, [&](size_t b, size_t e) {  // scale_tensor_to_0_1, vae.hpp:24-30
        for (size_t i = b; i < e; ++i) {
            const float v = (out_rgb[i] + 1.0f) * 0.5f;
            out_rgb[i]    = std::max(0.0f, std::min(1.0f, v));
        }
    });
    ctx->stats.last_decode_ms = now_ms() - t0;
    return true;
}

// ---- VAE encode (the data format in front of the path: pixels -> the init latent of img2img) -------------
static bool ensure_vae_encoder(sdm_ctx_t* ctx) {
    if (ctx->vae_enc_ready) return true;
    Runner& r        = ctx->vae_enc_runner;
    r.backend        = ctx->backend;
    r.ps.linear_type = GGML_TYPE_F16;
    r.graph_size     = 20480;
    ctx->vae_enc.init(r.ps, "first_stage_model.", ctx->vae.cfg);
    ctx->vae_enc.set_conv2d_scale(ctx->vae_conv2d_scale);  // AutoEncoderKL::set_conv2d_scale reaches every Conv2d of the autoencoder, encoder included
    if (!r.alloc_weights(ctx->params.weight_seed)) {
        set_error("VAE encoder weight buffer allocation failed");
        return false;
    }
    for (auto& sp : r.ps.specs) ctx->all_tensors.push_back({sp.name, sp.tensor});
    ctx->stats.weight_bytes += ggml_backend_buffer_get_size(r.weights);
    ctx->vae_enc_ready = true;
    return true;
}
// encode_first_stage (stable-diffusion.cpp:3042-3060) = VAE::encode (vae.hpp:112-168: x * 2 - 1, the encoder graph -> moments), gaussian_latent_sample
// (auto_encoder_kl.hpp:750-759: mean + exp(0.5 * clamp(logvar, -30, 20)) * randn from the context RNG — seeded with the request's seed, offset 0, stable-diffusion.cpp:5630),
// vae_to_diffusion_latents ((z - shift) * scale, :830-838).  rgb: planar f32 [w, h, 3, n] in [0, 1]; w, h multiples of 8; out: [w/8, h/8, zc, n].
// moments_out (optional, 2 * zc channels): what left the graph.
bool sd_vae_encode(sdm_ctx_t* ctx, const float* rgb, int w, int h, int n, uint64_t seed, float* out_latents, float* moments_out) {
    if (w < 8 || h < 8 || w % 8 || h % 8 || n < 1) {
        set_error("sd_vae_encode: width and height must be positive multiples of 8");
        return false;
    }
    if (!ensure_vae_encoder(ctx)) return false;
    Runner& r       = ctx->vae_enc_runner;
    const size_t ni = (size_t)w * h * 3 * n;
    std::vector<float> x(ni);
    for (size_t i = 0; i < ni; ++i) x[i] = rgb[i] * 2.0f - 1.0f;  // scale_tensor_to_minus1_1, vae.hpp:17-22
    const int zc = (int)ctx->vae.cfg.z_channels, lw = w / 8, lh = h / 8;
    auto build = [&](GraphCtx& g, std::vector<HostInput>& in) {
        g.flash_attn    = ctx->params.diffusion_flash_attn;
        g.conv_direct   = ctx->params.diffusion_conv_direct;
        ggml_tensor* tx = ggml_new_tensor_4d(g.ctx, GGML_TYPE_F32, w, h, 3, n);
        ggml_set_input(tx);
        in.push_back({tx, x.data(), ggml_nbytes(tx)});
        return ctx->vae_enc.forward(g, tx);
    };
    const size_t plane = (size_t)lw * lh, per = plane * zc;
    std::vector<float> moments(2 * per * n);
    char sig[64];
    snprintf(sig, sizeof(sig), "vae-enc %d %d %d s%g", w, h, n, (double)ctx->vae_conv2d_scale);
    if (!r.compute(build, moments.data(), moments.size() * sizeof(float), sig, {x.data()})) return false;
    if (moments_out) memcpy(moments_out, moments.data(), moments.size() * sizeof(float));
    PhiloxRNG rng(seed);
    const std::vector<float> noise = rng.randn((uint32_t)(per * n));  // randn_like(mean): one draw over the whole [lw, lh, zc, n] tensor
    const float sf = ctx->vae.cfg.scale_factor, sh = ctx->vae.cfg.shift_factor;
    for (int b = 0; b < n; ++b)
        for (size_t i = 0; i < per; ++i) {
            const float mean = moments[(size_t)b * 2 * per + i], logvar = moments[(size_t)b * 2 * per + per + i];
            const float stddev = std::exp(0.5f * std::max(-30.0f, std::min(20.0f, logvar)));
            const float z      = mean + stddev * noise[(size_t)b * per + i];
            out_latents[(size_t)b * per + i] = (z - sh) * sf;
        }
    return true;
}

// ---- TAESD decode ------------------------------------------------------
// TinyImageAutoEncoder (src/model/vae/tae.hpp:732-792; made at stable-diffusion.cpp:1407-1424 with the weights of `--taesd`, file prefix "tae."): parameters
// "tae.decoder.layers.<i>. ...", z_channels 16 for the DiT families.  Weights: synthetic like every other module until sd_load_weights_prefixed(ctx, file, "tae.") fills them.
static bool ensure_tae(sdm_ctx_t* ctx) {
    if (ctx->tae_ready) return true;
    Runner& r       = ctx->tae_runner;
    r.backend       = ctx->backend;
    r.ps.linear_type = GGML_TYPE_F16;
    r.graph_size    = 2048;
    ctx->tae.init(r.ps, "tae.decoder.layers.", ctx->is_dit ? 16 : 4);
    if (!r.alloc_weights(ctx->params.weight_seed)) {
        set_error("TAESD weight buffer allocation failed");
        return false;
    }
    for (auto& sp : r.ps.specs) ctx->all_tensors.push_back({sp.name, sp.tensor});
    ctx->stats.weight_bytes += ggml_backend_buffer_get_size(r.weights);
    ctx->tae_ready = true;
    return true;
}
// decode_first_stage with the tiny autoencoder: the diffusion latents enter as they are (diffusion_to_vae_latents is the identity, tae.hpp:763-765), the graph's output IS the
// image (scale_input = false, tae.hpp:745: no (x + 1) / 2, no clamp — the u8 conversion clamps); latents [w,h,zc,n] -> rgb f32 [8w,8h,3,n]
bool sd_tae_decode(sdm_ctx_t* ctx, const float* latents, int w, int h, int c, int n, float* out_rgb) {
    if (!ensure_tae(ctx)) return false;
    if (c != (int)ctx->tae.z_channels) {
        set_error("sd_tae_decode: the latents have " + std::to_string(c) + " channels, this model's TAESD takes " + std::to_string(ctx->tae.z_channels));
        return false;
    }
    Runner& r  = ctx->tae_runner;
    auto build = [&](GraphCtx& g, std::vector<HostInput>& in) {
        g.conv_direct   = ctx->params.diffusion_conv_direct;
        ggml_tensor* tz = ggml_new_tensor_4d(g.ctx, GGML_TYPE_F32, w, h, c, n);
        ggml_set_input(tz);
        in.push_back({tz, latents, ggml_nbytes(tz)});
        return ctx->tae.forward(g, tz);
    };
    const size_t on = (size_t)w * 8 * h * 8 * 3 * n;
    const double t0 = now_ms();
    char sig[64];
    snprintf(sig, sizeof(sig), "tae %d %d %d %d", w, h, c, n);
    if (!r.compute(build, out_rgb, on * sizeof(float), sig, {latents})) return false;
    ctx->stats.last_decode_ms = now_ms() - t0;
    return true;
}
// the reference decodes with TAESD instead of the KL-VAE when a taesd file is given and it is not for previews only (stable-diffusion.cpp:1496-1516): sdm_generate_image's switch
// sd_ctx_params_t::prediction (stable-diffusion.h prediction_t; resolved at stable-diffusion.cpp:1760-1790): v-prediction UNets (SD2.x 768-v, v-pred SDXL fine-tunes) use
// CompVisVDenoiser's scalings; the flow families have one parameterisation each
bool sd_set_prediction(sdm_ctx_t* ctx, int prediction) {
    if (prediction != SDM_EPS_PRED && prediction != SDM_V_PRED) {
        set_error("sd_set_prediction: only SDM_EPS_PRED and SDM_V_PRED are implemented");
        return false;
    }
    if (ctx->is_dit && prediction != SDM_EPS_PRED) {
        set_error("sd_set_prediction: the flow families (SD3.x, FLUX) have a fixed parameterisation");
        return false;
    }
    ctx->prediction = prediction;
    return true;
}
bool sd_use_tae(sdm_ctx_t* ctx, bool on) {
    if (on && !ensure_tae(ctx)) return false;
    ctx->tae_for_images = on;
    return true;
}

// ---- sample(): the denoise loop -------------------------------------------------------------------
// resolve_sample_method / sd_get_default_sample_method (stable-diffusion.cpp:3965-3975, 4006-4013): DiT families default to plain Euler
static int resolve_sample_method(const sdm_ctx_t* ctx, int m) {
    if (m == SDM_SAMPLE_METHOD_COUNT) return ctx->is_dit ? SDM_EULER_SAMPLE_METHOD : SDM_EULER_A_SAMPLE_METHOD;
    return m;
}
// resolve_scheduler / sd_get_default_scheduler (stable-diffusion.cpp:3977-3998, 4015-4022)
static int resolve_scheduler(const sdm_ctx_t* ctx, int scheduler, int method) {
    if (scheduler != SDM_SCHEDULER_COUNT) return scheduler;
    if (method == SDM_LCM_SAMPLE_METHOD || method == SDM_TCD_SAMPLE_METHOD) return SDM_LCM_SCHEDULER;
    if (method == SDM_DDIM_TRAILING_SAMPLE_METHOD) return SDM_SIMPLE_SCHEDULER;
    return ctx->is_flux ? SDM_FLUX_SCHEDULER : SDM_DISCRETE_SCHEDULER;
}
// method / scheduler / eta of one call, resolved as generate_image does (resolve_sample_method, resolve_scheduler, resolve_eta: stable-diffusion.cpp:4006-4049).  DDIM trailing IS
// Euler-A on the reference (sample_k_diffusion, denoiser.hpp:2843-2845) with its own eta / scheduler defaults: `method` comes back as Euler-A for it.
static bool resolve_sampling(const sdm_ctx_t* ctx, const sdm_sample_params_t& sp, int& method, int& scheduler, float& eta) {
    method    = resolve_sample_method(ctx, sp.sample_method);
    scheduler = resolve_scheduler(ctx, sp.scheduler, method);
    if (!sample_method_supported(method)) {
        set_error("sample method " + std::to_string(method) + " is not implemented (sdm_sample_method_t: 0 ... 20)");
        return false;
    }
    if (!scheduler_supported(scheduler)) {
        set_error("scheduler " + std::to_string(scheduler) + " is not implemented (sdm_scheduler_t)");
        return false;
    }
    eta = sp.eta == INFINITY ? default_eta(method) : sp.eta;
    if (method == SDM_DDIM_TRAILING_SAMPLE_METHOD) method = SDM_EULER_A_SAMPLE_METHOD;
    return true;
}
// scalings and timestep of one model call: Denoiser::get_scalings + sigma_to_t, and — sd_sample_params_t::shifted_timestep > 0 (timestep-shifted distilled UNets) —
// prepare_sample_timesteps / adjust_sample_step_scalings (stable-diffusion.cpp:2411-2457): the model sees round(t * shift / 1000) and the output scalings of THAT timestep's sigma
static void step_scalings(const sdm_ctx_t* ctx, const sdm_sample_params_t& sp, float sigma, float& c_skip, float& c_out, float& c_in, float& t) {
    ctx->scalings(sigma, c_skip, c_out, c_in);
    t = ctx->sigma_to_t(sigma);
    if (sp.shifted_timestep > 0 && !ctx->is_dit) {
        const float shifted_t_float = t * (float(sp.shifted_timestep) / float(TIMESTEPS));
        int64_t shifted_t           = static_cast<int64_t>(roundf(shifted_t_float));
        shifted_t                   = std::max((int64_t)0, std::min((int64_t)(TIMESTEPS - 1), shifted_t));
        t                           = (float)shifted_t;
        const int64_t idx           = static_cast<int64_t>(roundf(t));
        const float shifted_sigma   = ctx->denoiser.t_to_sigma((float)idx);
        float s_skip, s_out, s_in;
        ctx->scalings(shifted_sigma, s_skip, s_out, s_in);
        c_skip = s_skip * c_in / s_in;
        c_out  = s_out;
    }
}
// One denoiser call of the host loop — the reference's `denoise` lambda (stable-diffusion.cpp:2620-2926) on nb images: scalings and timestep of sigma, x * c_in, the model
// forward(s) (the cond / uncond pair in ONE graph when the conditionings allow it), classifier-free guidance, pred * c_out + x * c_skip.
struct HostDenoise {
    sdm_ctx_t* ctx;
    const sdm_img_gen_params_t* p;
    int W, H, C, nb;
    size_t per;
    bool use_cfg;
    std::vector<float> noised, cond_out, uncond_out, ts, skip_out;
    int n_sigmas = 0;  // sigmas.size() of the trajectory (GuidanceInput::schedule_size)
    std::vector<std::vector<float>> apg_momentum;  // adaptive projected guidance: one momentum buffer per image
    std::vector<float> x2, o2, t2, c2, y2;  // staging of the fused (cond, uncond) pair
    HostDenoise(sdm_ctx_t* ctx_, const sdm_img_gen_params_t* p_, int W_, int H_, int C_, int nb_)
        : ctx(ctx_), p(p_), W(W_), H(H_), C(C_), nb(nb_), per((size_t)W_ * H_ * C_), use_cfg(p_->sample_params.txt_cfg != 1.0f && p_->uncond.c_crossattn != nullptr),
          noised(per * nb_), cond_out(per * nb_), uncond_out(per * nb_), ts(nb_) {}
    // denoised_uncond (the CFG++ methods): GuiderOutput::pred_uncond = base_uncond * c_out + x * c_skip with the unconditional forward — or the conditional one when there is
    // no guidance pair (stable-diffusion.cpp:2877-2884)
    // step: the sampler's step index of this call (sample_k_diffusion hands the denoise callback i + 1, negated for the first stage of the two-stage methods): what
    // SkipLayerGuidance::is_enabled_for_step looks at (guidance.cpp:306-314)
    bool operator()(const float* x, float sigma, float* denoised, float* denoised_uncond = nullptr, int step = 0) {
        const sdm_sample_params_t& sp = p->sample_params;
        const size_t n = per * (size_t)nb;
        float c_skip, c_out, c_in, t;
        step_scalings(ctx, sp, sigma, c_skip, c_out, c_in, t);
        for (int b = 0; b < nb; ++b) ts[b] = t;
        for (size_t k = 0; k < n; ++k) noised[k] = x[k] * c_in;  // stable-diffusion.cpp:2662
        auto run = [&](const sd_condition_t& cd, float* dst) {
            return sd_unet_forward(ctx, noised.data(), W, H, C, nb, ts.data(), cd.c_crossattn, cd.ctx_dim, cd.n_tokens, 1,
                                   cd.c_vector, cd.vector_dim, 1, dst);
        };
        // the primary guidance of the (cond, uncond) pair: classifier-free (guidance.cpp:171) or, with any apg_* parameter set, adaptive projected guidance per image
        // (guidance.cpp:209-294; the momentum buffer lives as long as the trajectory, like the reference's guider object)
        auto guide = [&](float* dst) {
            const ApgParams apg{sp.apg_eta, sp.apg_momentum, sp.apg_norm_threshold, sp.apg_norm_threshold_smoothing};
            if (apg.enabled()) {
                if (apg_momentum.size() != (size_t)nb) apg_momentum.assign((size_t)nb, std::vector<float>());
                for (int b = 0; b < nb; ++b) apg_guided(&cond_out[(size_t)b * per], &uncond_out[(size_t)b * per], per, sp.txt_cfg, apg, apg_momentum[(size_t)b], dst + (size_t)b * per);
            } else {
                for (size_t k = 0; k < n; ++k) dst[k] = cfg_guided(cond_out[k], uncond_out[k], sp.txt_cfg);
            }
        };
        if (use_cfg && p->fuse_cfg_pair && p->cond.ctx_dim == p->uncond.ctx_dim && p->cond.n_tokens == p->uncond.n_tokens) {
            // one graph for the cond/uncond pair: images interleaved (b cond, b uncond, ...); context/y have batch 2 and are
            // tiled over the 2*nb images by the graph's ggml_repeat (dst[i] = src[i % 2])
            const size_t cn  = (size_t)p->cond.ctx_dim * p->cond.n_tokens;
            const bool has_y = p->cond.c_vector && p->uncond.c_vector;
            if (x2.empty()) {  // first call: the pair's conditioning is the same for every step, the staging buffers are reused
                x2.resize(2 * per * nb);
                o2.resize(2 * per * nb);
                t2.resize(2 * nb);
                c2.resize(2 * cn);
                memcpy(&c2[0], p->cond.c_crossattn, cn * sizeof(float));
                memcpy(&c2[cn], p->uncond.c_crossattn, cn * sizeof(float));
                if (has_y) {
                    y2.resize(2 * p->cond.vector_dim);
                    memcpy(&y2[0], p->cond.c_vector, p->cond.vector_dim * sizeof(float));
                    memcpy(&y2[p->cond.vector_dim], p->uncond.c_vector, p->cond.vector_dim * sizeof(float));
                }
            }
            std::fill(t2.begin(), t2.end(), t);
            for (int b = 0; b < nb; ++b) {
                memcpy(&x2[(2 * b) * per], &noised[b * per], per * sizeof(float));
                memcpy(&x2[(2 * b + 1) * per], &noised[b * per], per * sizeof(float));
            }
            if (!sd_unet_forward(ctx, x2.data(), W, H, C, 2 * nb, t2.data(), c2.data(), p->cond.ctx_dim, p->cond.n_tokens, 2,
                                 has_y ? y2.data() : nullptr, p->cond.vector_dim, 2, o2.data()))
                return false;
            for (int b = 0; b < nb; ++b) {
                memcpy(&cond_out[b * per], &o2[(2 * b) * per], per * sizeof(float));
                memcpy(&uncond_out[b * per], &o2[(2 * b + 1) * per], per * sizeof(float));
            }
            guide(denoised);
        } else if (!run(p->cond, cond_out.data())) {
            return false;
        } else if (use_cfg) {
            if (!run(p->uncond, uncond_out.data())) return false;
            guide(denoised);
        } else {
            for (size_t k = 0; k < n; ++k) denoised[k] = cond_out[k];
        }
        // skip-layer guidance (SkipLayerGuidance, guidance.cpp:296-340; stable-diffusion.cpp:2593-2611, 2860-2871): inside the step window one more conditional forward
        // WITHOUT the listed joint blocks; guided += (cond - skip) * scale.  DiT families only (the reference warns and ignores it elsewhere)
        if (ctx->is_dit && !ctx->is_flux && sp.slg_scale != 0.0f && sp.slg_layers && sp.slg_layer_count > 0) {
            const size_t schedule_size = (size_t)n_sigmas;
            const int start_step = static_cast<int>(sp.slg_layer_start * static_cast<float>(schedule_size));
            const int stop_step  = static_cast<int>(sp.slg_layer_end * static_cast<float>(schedule_size));
            if (schedule_size != 0 && step > start_step && step < stop_step) {
                const std::vector<int> skip(sp.slg_layers, sp.slg_layers + sp.slg_layer_count);
                skip_out.resize(n);
                if (!unet_forward_skip(ctx, noised.data(), W, H, C, nb, ts.data(), p->cond.c_crossattn, p->cond.ctx_dim, p->cond.n_tokens, 1, p->cond.c_vector, p->cond.vector_dim, 1,
                                       skip_out.data(), &skip))
                    return false;
                for (size_t k = 0; k < n; ++k) {
                    volatile float d = cond_out[k] - skip_out[k];  // (three separately rounded operations, like cfg_guided)
                    volatile float sc = d * sp.slg_scale;
                    denoised[k] += sc;
                }
            }
        }
        for (size_t k = 0; k < n; ++k) denoised[k] = denoised[k] * c_out + x[k] * c_skip;  // stable-diffusion.cpp:2876
        if (denoised_uncond) {
            const float* base = use_cfg ? uncond_out.data() : cond_out.data();
            for (size_t k = 0; k < n; ++k) denoised_uncond[k] = base[k] * c_out + x[k] * c_skip;
        }
        if (p->denoise_mask && p->init_latent) {  // inpainting: denoised * mask + init_latent * (1 - mask), the mask broadcast over channels and images (stable-diffusion.cpp:2888-2890)
            const size_t plane = (size_t)W * H;
            for (int b = 0; b < nb; ++b)
                for (int c = 0; c < C; ++c) {
                    float* d        = denoised + (size_t)b * per + (size_t)c * plane;
                    const float* il = p->init_latent + (size_t)c * plane;
                    for (size_t k = 0; k < plane; ++k) {
                        const float m = p->denoise_mask[k];
                        d[k]          = d[k] * m + il[k] * (1.0f - m);
                    }
                }
        }
        return true;
    }
};
// img2img (stable-diffusion.cpp:4924-4980): with an init latent and strength < 1 the trajectory starts t_enc = steps * strength steps before the end of the ladder
static std::vector<float> call_sigmas(sdm_ctx_t* ctx, const sdm_img_gen_params_t* p, int image_seq_len, int scheduler) {
    // set_flow_shift (stable-diffusion.cpp:3106-3115): the request's shift, or the family default (SD3.x 3.0, FLUX.1-dev 1.15), on the flow denoisers
    const float fs = p->sample_params.flow_shift;
    ctx->flow_denoiser.shift = (std::isfinite(fs) && fs > 0.f) ? fs : 3.0f;
    ctx->flux_denoiser.shift = (std::isfinite(fs) && fs > 0.f) ? fs : 1.15f;
    // custom sigmas replace the scheduler's ladder as they are (stable-diffusion.cpp:4337-4349)
    std::vector<float> sigmas = (p->sample_params.custom_sigmas && p->sample_params.custom_sigmas_count > 1)
                                    ? std::vector<float>(p->sample_params.custom_sigmas, p->sample_params.custom_sigmas + p->sample_params.custom_sigmas_count)
                                    : ctx->get_sigmas(p->sample_params.sample_steps, image_seq_len, scheduler);
    if (p->init_latent && p->strength < 1.f && !sigmas.empty()) {
        // (custom sigmas set sample_steps to their count - 1 first, stable-diffusion.cpp:4337-4345; a scheduler that returned fewer sigmas than steps + 1 — beta — likewise here)
        const int sample_steps = std::min(p->sample_params.sample_steps, (int)sigmas.size() - 1) < p->sample_params.sample_steps || p->sample_params.custom_sigmas_count > 1
                                     ? (int)sigmas.size() - 1
                                     : p->sample_params.sample_steps;
        size_t t_enc           = static_cast<size_t>(sample_steps * p->strength);
        if (t_enc == static_cast<size_t>(sample_steps)) t_enc--;
        const int64_t first = (int64_t)sample_steps - (int64_t)t_enc - 1;
        if (first > 0 && first < (int64_t)sigmas.size()) sigmas.erase(sigmas.begin(), sigmas.begin() + first);  // (a ladder shorter than steps + 1 — beta scheduler — keeps at least its last pair)
    }
    return sigmas;
}
// Denoiser::noise_scaling (denoiser.hpp:1181-1186 CompVis: latent + noise * sigma; :1274-1279 flow: latent * (1 - sigma) + noise * sigma); latent NULL = zeros (txt2img)
static inline void noise_scaling(float* x, const float* noise, const float* latent, size_t n, float sigma, bool flow) {
    if (!latent) {
        for (size_t i = 0; i < n; ++i) x[i] = 0.0f + noise[i] * sigma;
    } else if (flow) {
        const float om = 1.0f - sigma;
        for (size_t i = 0; i < n; ++i) x[i] = latent[i] * om + noise[i] * sigma;
    } else {
        for (size_t i = 0; i < n; ++i) x[i] = latent[i] + noise[i] * sigma;
    }
}
static bool sample_group(sdm_ctx_t* ctx, const sdm_img_gen_params_t* p, int b0, int nb, float* out) {
    const int W = p->width / 8, H = p->height / 8, C = ctx->in_channels();
    const size_t per = (size_t)W * H * C;
    const sdm_sample_params_t& sp = p->sample_params;
    int method, scheduler;
    float eta;
    if (!resolve_sampling(ctx, sp, method, scheduler, eta)) return false;
    const std::vector<float> sigmas = call_sigmas(ctx, p, W * H, scheduler);  // image_seq_len = latent pixels (stable-diffusion.cpp:2983-2986)
    const int steps                 = (int)sigmas.size() - 1;
    if (steps < 1) {
        set_error("the scheduler returned no sigma ladder for " + std::to_string(sp.sample_steps) + " steps");
        return false;
    }

    // per-image RNG: seed+b; initial noise consumes offset 0 (stable-diffusion.cpp:5678-5683; rng == sampler_rng :886-889)
    std::vector<PhiloxRNG> rngs;
    std::vector<float> x(per * nb);
    for (int b = 0; b < nb; ++b) rngs.emplace_back((uint64_t)(p->seed + b0 + b));
    parallel_chunks((size_t)nb, [&](size_t i0, size_t i1) {
        for (size_t b = i0; b < i1; ++b) {
            std::vector<float> noise = rngs[b].randn((uint32_t)per);
            noise_scaling(&x[b * per], noise.data(), p->init_latent, per, sigmas[0], ctx->is_dit);
        }
    }, 2);
    HostDenoise denoise(ctx, p, W, H, C, nb);
    denoise.n_sigmas = (int)sigmas.size();
    std::vector<float> denoised(per * nb);
    std::vector<std::vector<float>> step_noise(nb);

    if (method != SDM_EULER_SAMPLE_METHOD && method != SDM_EULER_A_SAMPLE_METHOD) {
        // the multi-stage / multi-step samplers (sampler.hpp: run_sampler_generic): per-image Philox streams, one draw of `per` normals per image and request
        const bool ok = run_sampler_generic(method, [&](const float* xin, float sigma, float* den, float* unc, int step) { return denoise(xin, sigma, den, unc, step); }, x, sigmas,
                                            [&](float* dst) {
                                                for (int b = 0; b < nb; ++b) {
                                                    const std::vector<float> nz = rngs[b].randn((uint32_t)per);
                                                    memcpy(dst + (size_t)b * per, nz.data(), per * sizeof(float));
                                                }
                                            },
                                            eta, ctx->is_dit, nb,
                                            [&](int b, uint32_t cnt, float* dst) {
                                                const std::vector<float> nz = rngs[b].randn(cnt);
                                                memcpy(dst, nz.data(), cnt * sizeof(float));
                                            });
        if (!ok) return false;
        memcpy(out, x.data(), x.size() * sizeof(float));
        return true;
    }

    for (int i = 0; i < steps; ++i) {
        const float sigma = sigmas[i], sigma_to = sigmas[i + 1];
        // The ancestral noise of this step depends on the sigma ladder only, not on the model output: draw it (host Philox, 0.7 ms per
        // SD1.5 image) on a helper thread WHILE the device runs the forward, instead of after it.  Same per-image streams, same order.
        float sigma_down = 0.f, sigma_up = 0.f, alpha_scale = 1.f;
        std::future<void> noise_job;
        if (method == SDM_EULER_A_SAMPLE_METHOD && sigma_to != 0.f && eta != 0.f) {
            if (ctx->is_dit)  // flow denoisers (SD3.5, FLUX): get_ancestral_step(..., is_flow_denoiser), denoiser.hpp:1501-1511
                ancestral_step_flow(sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale);
            else
                ancestral_step(sigma, sigma_to, eta, sigma_down, sigma_up);
            if (sigma_up > 0.f)
                noise_job = std::async(std::launch::async, [&]() {
                    for (int b = 0; b < nb; ++b) step_noise[b] = rngs[b].randn((uint32_t)per);
                });
        }
        struct JoinNoise {  // an early return must not leave the helper writing into dead stack frames
            std::future<void>& f;
            ~JoinNoise() {
                if (f.valid()) f.wait();
            }
        } join_noise{noise_job};
        if (!denoise(x.data(), sigma, denoised.data(), nullptr, i + 1)) return false;
        // the update itself: sample_euler_ancestral / sample_euler (sampler.hpp: sampler_update — the function tests hold bit-for-bit against the reference's
        // own src/runtime/denoiser.hpp compiled into oracle/_ref)
        sampler_update(x.data(), denoised.data(), per, nb, method == SDM_EULER_A_SAMPLE_METHOD, ctx->is_dit, sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale,
                       [&](int b) -> const float* {
                           if (noise_job.valid()) noise_job.get();
                           return step_noise[b].data();
                       });
    }
    memcpy(out, x.data(), x.size() * sizeof(float));
    return true;
}

// ---- device-resident sampler -----------------------------------------------------------------------
// The reference crosses the host boundary three times per model call (x*c_in up, eps down, CFG / Euler on the host:
// stable-diffusion.cpp:2636-2664, 2855-2896; denoiser.hpp:1513-1546).  Here one graph per step carries the whole iteration —
//   x*c_in -> (cond, uncond interleaved) model call -> uncond + s*(cond - uncond) -> *c_out + x*c_skip -> Euler(-A) update (+ noise*sigma_up)
//   -> CPY back into the persistent latent tensor
// — on nodes the backend already runs (MUL / SUB / ADD / DIV with a 1-element broadcast operand, REPEAT, CPY), and every step's
// scalars arrive as ONE 8-float input, so all steps share one cached plan.  Nothing is read back until the last step: uploads and
// graphs are queued on the backend stream (set_tensor_async / graph_compute_async) and the host builds step k+1 while step k runs.
// Ancestral noise stays the host Philox stream (bit-reproducible, rng_philox.hpp:101-122), uploaded per step (64 KB per SD1.5 image).
static bool sample_group_device(sdm_ctx_t* ctx, const sdm_img_gen_params_t* p, int b0, int nb, float* out, bool* handled) {
    *handled = false;
    const int W = p->width / 8, H = p->height / 8, C = ctx->in_channels();
    const sdm_sample_params_t& sp = p->sample_params;
    const bool use_cfg = sp.txt_cfg != 1.0f && p->uncond.c_crossattn != nullptr;
    const bool has_y   = p->cond.c_vector != nullptr;
    if (use_cfg && (p->cond.ctx_dim != p->uncond.ctx_dim || p->cond.n_tokens != p->uncond.n_tokens || has_y != (p->uncond.c_vector != nullptr))) return true;
    if (ctx->out_channels() != C) return true;  // learned-sigma heads are not sampled this way
    int method, scheduler;
    float eta;
    if (!resolve_sampling(ctx, sp, method, scheduler, eta)) {
        *handled = true;  // (the error is set: the host loop would refuse the same parameters)
        return false;
    }
    if (method != SDM_EULER_SAMPLE_METHOD && method != SDM_EULER_A_SAMPLE_METHOD) return true;  // multi-stage / multi-step samplers: the host loop around the device forward
    if (p->denoise_mask && p->init_latent) return true;  // the inpainting blend lives in the host loop's denoise call
    if (ApgParams{sp.apg_eta, sp.apg_momentum, sp.apg_norm_threshold, sp.apg_norm_threshold_smoothing}.enabled()) return true;  // adaptive projected guidance: norms and a momentum buffer on the host
    if (sp.slg_scale != 0.0f && sp.slg_layers && sp.slg_layer_count > 0 && ctx->is_dit && !ctx->is_flux) return true;  // skip-layer guidance: a third forward inside a step window
    *handled = true;
    const size_t per = (size_t)W * H * C;
    const std::vector<float> sigmas = call_sigmas(ctx, p, W * H, scheduler);
    const int steps                 = (int)sigmas.size() - 1;
    if (steps < 1) {
        set_error("the scheduler returned no sigma ladder for " + std::to_string(sp.sample_steps) + " steps");
        return false;
    }
    const bool euler_a              = method == SDM_EULER_A_SAMPLE_METHOD;
    const bool flow                 = ctx->is_dit;  // flow denoiser: ancestral steps rescale x by alpha_scale before the noise (denoiser.hpp:1536-1541)

    if (!ctx->sstate || ctx->sstate->W != W || ctx->sstate->H != H || ctx->sstate->C != C || ctx->sstate->N != nb) {
        ctx->sstate.reset(new sdm_ctx_t::SamplerState());
        auto& st = *ctx->sstate;
        ggml_init_params ip{0, nullptr, true};
        st.sctx  = ggml_init(ip);
        st.x     = ggml_new_tensor_4d(st.sctx, GGML_TYPE_F32, W, H, C, nb);
        st.noise = ggml_new_tensor_4d(st.sctx, GGML_TYPE_F32, W, H, C, nb);
        st.eps   = ggml_new_tensor_4d(st.sctx, GGML_TYPE_F32, W, H, C, nb);  // CFG-pair split: this rank's weighted eps, summed in place by the exchange
        ggml_set_name(st.eps, "sampler.eps");
        ggml_set_name(st.x, "sampler.x");
        ggml_set_name(st.noise, "sampler.noise");
        st.buf = ggml_backend_alloc_ctx_tensors(st.sctx, ctx->backend);
        if (!st.buf) {
            ctx->sstate.reset();
            set_error("sampler state allocation failed");
            return false;
        }
        st.W = W, st.H = H, st.C = C, st.N = nb;
    }
    auto& st = *ctx->sstate;

    std::vector<PhiloxRNG> rngs;
    std::vector<float> x(per * nb), noise(per * nb, 0.f);
    for (int b = 0; b < nb; ++b) rngs.emplace_back((uint64_t)(p->seed + b0 + b));
    // one independent Philox stream per image (seed + b): the images' draws run on separate host threads (0.7 ms per SD1.5 image each)
    parallel_chunks((size_t)nb, [&](size_t i0, size_t i1) {
        for (size_t b = i0; b < i1; ++b) {
            std::vector<float> nz = rngs[b].randn((uint32_t)per);
            noise_scaling(&x[b * per], nz.data(), p->init_latent, per, sigmas[0], ctx->is_dit);
        }
    }, 2);
    ggml_backend_tensor_set_async(ctx->backend, st.x, x.data(), 0, x.size() * sizeof(float));
    ggml_backend_tensor_set_async(ctx->backend, st.noise, noise.data(), 0, noise.size() * sizeof(float));

    // CFG-pair split (guidance.cpp:149-179 computed on two GPUs): this context runs ONE branch per step and writes
    // weight * eps (weight = s on the cond rank, 1 - s on the uncond rank) to st.eps; the caller's exchange sums the two ranks' buffers in
    // place, in HBM, ordered on the backend stream; a second small graph takes the Euler(-A) update from the sum.  Both ranks then hold the
    // same latents (same Philox streams), so nothing else is exchanged.
    const bool pair = ctx->pair_fn != nullptr && use_cfg;
    // conditioning of the (cond, uncond) pair, tiled over the images by the model graph's own ggml_repeat
    const bool both   = use_cfg && !pair;
    const int n_model = both ? 2 * nb : nb;
    const int ctx_n   = both ? 2 : 1;
    const size_t cn   = (size_t)p->cond.ctx_dim * p->cond.n_tokens;
    const sd_condition_t& first = (pair && ctx->pair_branch == 1) ? p->uncond : p->cond;
    std::vector<float> c2(cn * ctx_n), y2;
    memcpy(&c2[0], first.c_crossattn, cn * sizeof(float));
    if (both) memcpy(&c2[cn], p->uncond.c_crossattn, cn * sizeof(float));
    if (has_y) {
        y2.resize((size_t)p->cond.vector_dim * ctx_n);
        memcpy(&y2[0], first.c_vector, p->cond.vector_dim * sizeof(float));
        if (both) memcpy(&y2[p->cond.vector_dim], p->uncond.c_vector, p->cond.vector_dim * sizeof(float));
    }
    void* pair_stream = nullptr;
    if (pair) {
        typedef void* (*get_stream_fn)(ggml_backend_t);
        ggml_backend_dev_t dev = ggml_backend_get_device(ctx->backend);
        ggml_backend_reg_t reg = dev ? ggml_backend_dev_backend_reg(dev) : nullptr;
        get_stream_fn gs       = reg ? (get_stream_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_get_stream") : nullptr;
        pair_stream            = gs ? gs(ctx->backend) : nullptr;  // host backends (the CPU oracle in the tests): no stream, calls are synchronous
    }
    ModelSideInputs si;
    if (!prepare_side_inputs(ctx, W, H, n_model, p->cond.n_tokens, has_y, si)) return false;
    std::vector<float> ts(n_model);
    Runner& r = ctx->unet_runner;
    char step_sig[200];  // every step of the trajectory replays ONE cached graph: only the 8 scalars, the timesteps and the conditioning are re-uploaded
    snprintf(step_sig, sizeof(step_sig), "step %d %d %d %d cfg%d ea%d %lld %lld y%lld %p pair%d", W, H, C, nb, (int)use_cfg, (int)euler_a + 2 * (int)flow, (long long)p->cond.ctx_dim,
             (long long)p->cond.n_tokens, (long long)(has_y ? p->cond.vector_dim : -1), (void*)st.x, (int)pair);
    const std::string update_sig = std::string("update ") + step_sig;

    for (int i = 0; i < steps; ++i) {
        const float sigma = sigmas[i], sigma_to = sigmas[i + 1];
        float c_skip, c_out, c_in, t_model;
        step_scalings(ctx, sp, sigma, c_skip, c_out, c_in, t_model);
        std::fill(ts.begin(), ts.end(), t_model);
        // scalars of this step: {c_in, cfg scale, c_out, c_skip, a, b, noise gain, alpha}; Euler-A: x' = (a*x + b*denoised) [* alpha on flow models] + gain*noise;
        // Euler: x' = x + ((x - denoised) / a) * b with a = sigma, b = sigma_to - sigma
        float sc[8] = {c_in, pair ? (ctx->pair_branch == 0 ? sp.txt_cfg : 1.0f - sp.txt_cfg) : sp.txt_cfg, c_out, c_skip, 0.f, 0.f, 0.f, 1.f};
        bool fresh_noise = false;
        if (euler_a) {
            if (sigma_to == 0.f) {
                sc[4] = 0.f, sc[5] = 1.f;
            } else if (eta == 0.f) {
                const float ratio = sigma_to / sigma;
                sc[4] = ratio, sc[5] = (float)(1.0 - ratio);
            } else {
                float sigma_down, sigma_up, alpha_scale = 1.f;
                if (flow)
                    ancestral_step_flow(sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale);
                else
                    ancestral_step(sigma, sigma_to, eta, sigma_down, sigma_up);
                const float ratio = sigma_down / sigma;
                sc[4] = ratio, sc[5] = 1.0f - ratio;
                if (sigma_up > 0.f) {
                    sc[6]       = sigma_up;
                    sc[7]       = alpha_scale;
                    fresh_noise = true;
                }
            }
        } else {
            sc[4] = sigma, sc[5] = sigma_to - sigma;
        }
        if (fresh_noise) {
            parallel_chunks((size_t)nb, [&](size_t i0, size_t i1) {
                for (size_t b = i0; b < i1; ++b) {
                    std::vector<float> nz = rngs[b].randn((uint32_t)per);
                    memcpy(&noise[b * per], nz.data(), per * sizeof(float));
                }
            }, 2);
            ggml_backend_tensor_set_async(ctx->backend, st.noise, noise.data(), 0, noise.size() * sizeof(float));
        }
        auto build = [&](GraphCtx& g, std::vector<HostInput>& in) {
            ggml_context* c  = g.ctx;
            ggml_tensor* tsc = ggml_new_tensor_1d(c, GGML_TYPE_F32, 8);
            ggml_set_input(tsc);
            in.push_back({tsc, sc, sizeof(sc)});
            auto S = [&](int k) { return ggml_view_1d(c, tsc, 1, (size_t)k * sizeof(float)); };
            ggml_tensor* xs     = st.x;
            ggml_tensor* noised = ggml_mul(c, xs, S(0));
            ggml_tensor* xin    = noised;
            if (both) {  // every image twice, (cond, uncond) adjacent: [per, 1, nb] -> [per, 2, nb]
                ggml_tensor* flat = ggml_reshape_3d(c, noised, (int64_t)per, 1, nb);
                ggml_tensor* rep  = ggml_repeat(c, flat, ggml_new_tensor_3d(c, GGML_TYPE_F32, (int64_t)per, 2, nb));
                xin               = ggml_reshape_4d(c, rep, W, H, C, 2 * nb);
            }
            ggml_tensor* eps = build_model_call(ctx, g, in, xin, n_model, ts.data(), c2.data(), p->cond.ctx_dim, p->cond.n_tokens, ctx_n,
                                                has_y ? y2.data() : nullptr, p->cond.vector_dim, ctx_n, si);
            if (pair) return ggml_cpy(c, ggml_mul(c, eps, S(1)), st.eps);  // weight * eps of this rank's branch; the update follows the exchange
            ggml_tensor* guided = eps;
            if (both) {  // uncond + s*(cond - uncond), guidance.cpp:171
                ggml_tensor* e3 = ggml_reshape_3d(c, ggml_cont(c, eps), (int64_t)per, 2, nb);
                ggml_tensor* ec = ggml_view_3d(c, e3, (int64_t)per, 1, nb, e3->nb[1], e3->nb[2], 0);
                ggml_tensor* eu = ggml_view_3d(c, e3, (int64_t)per, 1, nb, e3->nb[1], e3->nb[2], e3->nb[1]);
                ggml_tensor* d  = ggml_mul(c, ggml_sub(c, ec, eu), S(1));
                guided          = ggml_reshape_4d(c, ggml_add(c, eu, d), W, H, C, nb);
            }
            ggml_tensor* den = ggml_add(c, ggml_mul(c, guided, S(2)), ggml_mul(c, xs, S(3)));  // stable-diffusion.cpp:2876
            ggml_tensor* xn;
            if (euler_a) {  // denoiser.hpp:1513-1546
                xn = ggml_add(c, ggml_mul(c, xs, S(4)), ggml_mul(c, den, S(5)));
                if (flow) xn = ggml_mul(c, xn, S(7));
                xn = ggml_add(c, xn, ggml_mul(c, st.noise, S(6)));
            } else {  // denoiser.hpp:1582-1597
                ggml_tensor* d = ggml_div(c, ggml_sub(c, xs, den), S(4));
                xn             = ggml_add(c, xs, ggml_mul(c, d, S(5)));
            }
            return ggml_cpy(c, xn, xs);
        };
        std::vector<const void*> ptrs{sc, ts.data(), c2.data()};
        if (has_y) ptrs.push_back(y2.data());
        if (ctx->is_flux) {
            ptrs.push_back(si.guidance.data());
            ptrs.push_back(si.pe().data());
        }
        if (!r.compute(build, nullptr, 0, step_sig, ptrs)) return false;
        if (pair) {
            if (!ctx->pair_fn(st.eps->data, (int64_t)(per * nb), pair_stream, ctx->pair_user)) {
                set_error("CFG-pair exchange failed");
                return false;
            }
            auto update = [&](GraphCtx& g, std::vector<HostInput>& in) {  // s*cond + (1-s)*uncond = uncond + s*(cond - uncond) is in st.eps
                ggml_context* c  = g.ctx;
                ggml_tensor* tsc = ggml_new_tensor_1d(c, GGML_TYPE_F32, 8);
                ggml_set_input(tsc);
                in.push_back({tsc, sc, sizeof(sc)});
                auto S = [&](int k) { return ggml_view_1d(c, tsc, 1, (size_t)k * sizeof(float)); };
                ggml_tensor* xs  = st.x;
                ggml_tensor* den = ggml_add(c, ggml_mul(c, st.eps, S(2)), ggml_mul(c, xs, S(3)));
                ggml_tensor* xn;
                if (euler_a) {
                    xn = ggml_add(c, ggml_mul(c, xs, S(4)), ggml_mul(c, den, S(5)));
                    if (flow) xn = ggml_mul(c, xn, S(7));
                    xn = ggml_add(c, xn, ggml_mul(c, st.noise, S(6)));
                } else {
                    ggml_tensor* d = ggml_div(c, ggml_sub(c, xs, den), S(4));
                    xn             = ggml_add(c, xs, ggml_mul(c, d, S(5)));
                }
                return ggml_cpy(c, xn, xs);
            };
            if (!ctx->pair_runner.compute(update, nullptr, 0, update_sig, {sc})) return false;
        }
    }
    ggml_backend_tensor_get(st.x, out, 0, per * nb * sizeof(float));  // synchronises the stream
    ctx->stats.unet_calls  = r.calls;
    ctx->stats.graph_nodes = r.last_nodes;
    if (r.galloc) ctx->stats.compute_buffer_bytes = ggml_gallocr_get_buffer_size(r.galloc, 0);
    return true;
}

bool sd_sample_latents(sdm_ctx_t* ctx, const sdm_img_gen_params_t* p, float* out_latents) {
    const int W = p->width / 8, H = p->height / 8, C = ctx->in_channels();
    const size_t per = (size_t)W * H * C;
    const int group  = p->device_batch > 0 ? p->device_batch : p->batch_count;
    const double t0  = now_ms();
    for (int b0 = 0; b0 < p->batch_count; b0 += group) {
        const int nb = std::min(group, p->batch_count - b0);
        if (p->device_sampler) {
            bool handled = false;
            if (!sample_group_device(ctx, p, b0, nb, out_latents + b0 * per, &handled)) return false;
            if (handled) continue;  // otherwise (cond / uncond shapes differ): the host loop below
        }
        if (!sample_group(ctx, p, b0, nb, out_latents + b0 * per)) return false;
    }
    ctx->stats.last_sample_ms = now_ms() - t0;
    return true;
}

static inline uint8_t float_to_u8(float v) {  // preprocessing.hpp:27-35
    if (v <= 0.0f) return 0;
    if (v >= 1.0f) return 255;
    return (uint8_t)(v * 255.0f + 0.5f);
}

// planar CHW floats in [0, 1] -> interleaved RGB bytes (preprocessing_tensor_frame_to_sd_image, src/runtime/preprocessing.hpp:37-60)
static void planar_rgb_to_u8(const float* f, size_t pix, uint8_t* d) {
    parallel_chunks(pix, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            d[i * 3 + 0] = float_to_u8(f[i]);
            d[i * 3 + 1] = float_to_u8(f[pix + i]);
            d[i * 3 + 2] = float_to_u8(f[2 * pix + i]);
        }
    });
}
// the same conversion on caller memory — tests only (bit-exact against the reference's preprocessing.hpp compiled into oracle/_ref)
void sd_planar_rgb_to_u8(const float* chw, int width, int height, uint8_t* out) { planar_rgb_to_u8(chw, (size_t)width * height, out); }

bool sdm_generate_image(sdm_ctx_t* ctx, const sdm_img_gen_params_t* p, sdm_image_t** images_out, int* num_images_out) {
    const int W = p->width / 8, H = p->height / 8, C = ctx->in_channels();
    const size_t per = (size_t)W * H * C;
    std::vector<float> latents(per * p->batch_count);
    if (!sd_sample_latents(ctx, p, latents.data())) return false;
    const int PW = W * 8, PH = H * 8;
    const size_t pix = (size_t)PW * PH;
    sdm_image_t* imgs = (sdm_image_t*)calloc(p->batch_count, sizeof(sdm_image_t));  // stable-diffusion.cpp:5398-5410
    std::vector<float> rgb(pix * 3 * p->batch_count);
    const int group = p->device_batch > 0 ? p->device_batch : p->batch_count;
    double dec_ms   = 0;
    for (int b0 = 0; b0 < p->batch_count; b0 += group) {
        const int nb = std::min(group, p->batch_count - b0);
        if (!(ctx->tae_for_images ? sd_tae_decode : sd_vae_decode)(ctx, latents.data() + b0 * per, W, H, C, nb, rgb.data() + (size_t)b0 * pix * 3)) {
            sdm_free_images(imgs, p->batch_count);
            return false;
        }
        dec_ms += ctx->stats.last_decode_ms;
    }
    ctx->stats.last_decode_ms = dec_ms;
    for (int b = 0; b < p->batch_count; ++b) {
        imgs[b].width   = PW;
        imgs[b].height  = PH;
        imgs[b].channel = 3;
        imgs[b].data    = (uint8_t*)malloc(pix * 3);
        const float* f  = rgb.data() + (size_t)b * pix * 3;
        uint8_t* d      = imgs[b].data;
        planar_rgb_to_u8(f, pix, d);
    }
    *images_out     = imgs;
    *num_images_out = p->batch_count;
    return true;
}

void sdm_free_images(sdm_image_t* images, int num_images) {
    if (!images) return;
    for (int i = 0; i < num_images; ++i) free(images[i].data);
    free(images);
}

void sd_philox_randn(uint64_t seed, uint32_t offset, uint32_t n, float* out) {
    PhiloxRNG r(seed);
    r.offset = offset;
    std::vector<float> v = r.randn(n);
    memcpy(out, v.data(), n * sizeof(float));
}
void sd_philox_uint32(uint64_t seed, uint32_t offset, uint32_t n, uint32_t* out) {
    for (uint32_t i = 0; i < n; ++i) PhiloxRNG::words(seed, offset, i, out + 4 * (size_t)i);
}
int sd_get_sigmas(int steps, float* out) {
    static CompVisDenoiser d;
    std::vector<float> s = d.get_sigmas(steps);
    memcpy(out, s.data(), s.size() * sizeof(float));
    return (int)s.size();
}
void sd_set_guidance(sdm_ctx_t* ctx, float guidance) { ctx->guidance = guidance; }
// the host sampler's CFG combine on n floats (cfg_guided, sampler.hpp) — exported so that tests can hold it bit-for-bit against the reference's own
// ClassifierFreeGuidance::forward compiled into oracle/_ref (src/runtime/guidance.cpp:149-179)
// The host sampler's loop — sigma ladder, initial noise, per-step scalings / timestep, ancestral step, update, Philox noise order — on ONE image of n floats
// with a SYNTHETIC model in place of the network: denoised = x * (1 / (1 + sigma)) + 0.01 * sigma  (plain f32 arithmetic the reference-side wrapper
// oracle/ref_denoiser_wrap.cpp states identically).  family: 0 = CompVis (SD1.x / SDXL), 1 = discrete flow (SD3.x, shift 3), 2 = FLUX flow; method: the
// sdm_sample_method_t value; eta: INFINITY = the method's default.  aux (optional, 5 floats per step): c_skip, c_out, c_in, t, sigma — what the loop fed the model.
// Tests only (tests/test_host_logic.py: bit-for-bit against the reference's src/runtime/denoiser.hpp compiled into oracle/_ref).
int sd_sample_synthetic(int family, int steps, int image_seq_len, int64_t n, uint64_t seed, int method, float eta, float* out, float* aux) {
    if (steps < 1 || n < 1 || family < 0 || family > 2) return -1;
    CompVisDenoiser cv;
    DiscreteFlowDenoiser fl;
    FluxFlowDenoiser fx;
    const bool flow = family != 0;
    if (eta == INFINITY) eta = method == SDM_EULER_A_SAMPLE_METHOD ? 1.0f : 0.0f;
    const std::vector<float> sigmas = family == 0 ? cv.get_sigmas((uint32_t)steps) : (family == 1 ? fl.get_sigmas((uint32_t)steps) : fx.get_sigmas((uint32_t)steps, image_seq_len));
    PhiloxRNG rng(seed);
    std::vector<float> x = rng.randn((uint32_t)n), den((size_t)n), nz;
    for (int64_t k = 0; k < n; ++k) x[k] = 0.0f + x[k] * sigmas[0];
    for (int i = 0; i + 1 < (int)sigmas.size(); ++i) {
        const float sigma = sigmas[i], sigma_to = sigmas[i + 1];
        float c_skip, c_out, c_in;
        if (family == 0) cv.scalings(sigma, c_skip, c_out, c_in);
        else if (family == 1) fl.scalings(sigma, c_skip, c_out, c_in);
        else fx.scalings(sigma, c_skip, c_out, c_in);
        const float t = family == 0 ? cv.sigma_to_t(sigma) : (family == 1 ? fl.sigma_to_t(sigma) : sigma);
        if (aux) {
            float* a = aux + 5 * i;
            a[0] = c_skip, a[1] = c_out, a[2] = c_in, a[3] = t, a[4] = sigma;
        }
        const float g = 1.0f / (1.0f + sigma), h = 0.01f * sigma;
        for (int64_t k = 0; k < n; ++k) den[k] = x[k] * g + h;
        float sigma_down = 0.f, sigma_up = 0.f, alpha_scale = 1.f;
        if (method == SDM_EULER_A_SAMPLE_METHOD && sigma_to != 0.f && eta != 0.f) {
            if (flow) ancestral_step_flow(sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale);
            else ancestral_step(sigma, sigma_to, eta, sigma_down, sigma_up);
        }
        sampler_update(x.data(), den.data(), (size_t)n, 1, method == SDM_EULER_A_SAMPLE_METHOD, flow, sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale,
                       [&](int) -> const float* {
                           nz = rng.randn((uint32_t)n);
                           return nz.data();
                       });
    }
    memcpy(out, x.data(), sizeof(float) * (size_t)n);
    return (int)sigmas.size();
}
// sd_sample_synthetic with a scheduler and every implemented method (the loop sample_group runs, on the synthetic model); aux: 5 floats per model call, in call order
int sd_sample_synthetic2(int family, int steps, int image_seq_len, int64_t n, uint64_t seed, int method, int scheduler, float eta, float* out, float* aux, int aux_calls) {
    if (steps < 1 || n < 1 || family < 0 || family > 4 || !sample_method_supported(method)) return -1;  // family 4: CompVis with the v-prediction scalings
    CompVisDenoiser cv;
    DiscreteFlowDenoiser fl;
    FluxFlowDenoiser fx;
    const int fam   = (family == 3 || family == 4) ? 0 : family;
    const bool flow = fam != 0;
    if (scheduler == SDM_SCHEDULER_COUNT) scheduler = (method == SM_LCM || method == SM_TCD) ? SCHED_LCM : (method == SM_DDIM_TRAILING ? SCHED_SIMPLE : (fam == 2 ? SCHED_FLUX : SCHED_DISCRETE));
    if (!scheduler_supported(scheduler)) return -1;
    if (eta == INFINITY) eta = default_eta(method);
    if (method == SM_DDIM_TRAILING) method = SM_EULER_A;
    std::vector<float> sigmas((size_t)steps + 2);
    const int ns = sd_get_sigmas_sched(family, scheduler, steps, image_seq_len, 0.f, sigmas.data());
    if (ns < 2) return -1;
    sigmas.resize((size_t)ns);
    PhiloxRNG rng(seed);
    std::vector<float> x = rng.randn((uint32_t)n);
    for (int64_t k = 0; k < n; ++k) x[k] = 0.0f + x[k] * sigmas[0];
    int calls  = 0;
    auto model = [&](const float* xin, float sigma, float* den, float* unc, int /*step*/) {
        float c_skip, c_out, c_in;
        if (family == 4) {  // CompVisVDenoiser::get_scalings, sigma_data = 1 (the expressions sdm_ctx_t::scalings uses for SDM_V_PRED)
            const float sigma_data = 1.0f;
            c_skip = sigma_data * sigma_data / (sigma * sigma + sigma_data * sigma_data);
            c_out  = -sigma * sigma_data / std::sqrt(sigma * sigma + sigma_data * sigma_data);
            c_in   = 1.0f / std::sqrt(sigma * sigma + sigma_data * sigma_data);
        } else if (fam == 0) cv.scalings(sigma, c_skip, c_out, c_in);
        else if (fam == 1) fl.scalings(sigma, c_skip, c_out, c_in);
        else fx.scalings(sigma, c_skip, c_out, c_in);
        const float t = fam == 0 ? cv.sigma_to_t(sigma) : (fam == 1 ? fl.sigma_to_t(sigma) : sigma);
        if (aux && calls < aux_calls) {
            float* a = aux + 5 * (size_t)calls;
            a[0] = c_skip, a[1] = c_out, a[2] = c_in, a[3] = t, a[4] = sigma;
        }
        ++calls;
        const float g = 1.0f / (1.0f + sigma), h = 0.01f * sigma;
        for (int64_t k = 0; k < n; ++k) den[k] = xin[k] * g + h;
        if (unc) {  // the synthetic model's unconditional prediction (the CFG++ methods)
            const float gu = 0.9f / (1.0f + sigma), hu = -0.02f * sigma;
            for (int64_t k = 0; k < n; ++k) unc[k] = xin[k] * gu + hu;
        }
        return true;
    };
    if (method == SM_EULER || method == SM_EULER_A) {
        std::vector<float> den((size_t)n), nz;
        for (int i = 0; i + 1 < (int)sigmas.size(); ++i) {
            const float sigma = sigmas[i], sigma_to = sigmas[i + 1];
            model(x.data(), sigma, den.data(), nullptr, i + 1);
            float sigma_down = 0.f, sigma_up = 0.f, alpha_scale = 1.f;
            if (method == SM_EULER_A && sigma_to != 0.f && eta != 0.f) {
                if (flow) ancestral_step_flow(sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale);
                else ancestral_step(sigma, sigma_to, eta, sigma_down, sigma_up);
            }
            sampler_update(x.data(), den.data(), (size_t)n, 1, method == SM_EULER_A, flow, sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale, [&](int) -> const float* {
                nz = rng.randn((uint32_t)n);
                return nz.data();
            });
        }
    } else if (!run_sampler_generic(method, model, x, sigmas, [&](float* dst) {
                   const std::vector<float> nz = rng.randn((uint32_t)n);
                   memcpy(dst, nz.data(), sizeof(float) * (size_t)n);
               }, eta, flow, 1, [&](int, uint32_t cnt, float* dst) {
                   const std::vector<float> nz = rng.randn(cnt);
                   memcpy(dst, nz.data(), sizeof(float) * cnt);
               })) {
        return -1;
    }
    memcpy(out, x.data(), sizeof(float) * (size_t)n);
    return calls;
}
int sd_get_sigmas_sched(int family, int scheduler, int steps, int image_seq_len, float shift, float* out) {
    if (family < 0 || family > 4 || steps < 0 || !scheduler_supported(scheduler)) return -1;
    std::vector<float> s;
    if (scheduler == SCHED_FLUX) {
        FluxFlowDenoiser d;
        s = d.get_sigmas((uint32_t)steps, image_seq_len);
    } else if (family == 0 || family == 3 || family == 4) {
        static const CompVisDenoiser d;
        s = scheduler_sigmas(scheduler, (uint32_t)steps, d.sigma_min(), d.sigma_max(), [&](float t) { return d.t_to_sigma(t); }, family == 3 ? 1 : 0);
    } else if (family == 1) {
        DiscreteFlowDenoiser d;
        if (shift > 0.f) d.shift = shift;
        s = scheduler_sigmas(scheduler, (uint32_t)steps, d.sigma_min(), d.sigma_max(), [&](float t) { return d.t_to_sigma(t); }, -1);
    } else {
        FluxFlowDenoiser d;
        if (shift > 0.f) d.shift = shift;
        s = scheduler_sigmas(scheduler, (uint32_t)steps, d.sigma_min(), d.sigma_max(), [&](float t) { return d.t_to_sigma(t); }, -1);
    }
    memcpy(out, s.data(), s.size() * sizeof(float));
    return (int)s.size();
}
// adaptive projected guidance over `steps` successive denoise calls of one image (momentum carried), for the bit-exact test against the reference's guidance.cpp
void sd_apg_sequence(const float* cond, const float* uncond, int64_t n, int steps, float scale, float eta, float momentum, float norm_threshold, float norm_threshold_smoothing, float* out) {
    const ApgParams prm{eta, momentum, norm_threshold, norm_threshold_smoothing};
    std::vector<float> buf;
    for (int s = 0; s < steps; ++s) apg_guided(cond + (size_t)s * n, uncond + (size_t)s * n, (size_t)n, scale, prm, buf, out + (size_t)s * n);
}
void sd_cfg_combine(const float* cond, const float* uncond, int64_t n, float scale, float* out) {
    for (int64_t k = 0; k < n; ++k) out[k] = cfg_guided(cond[k], uncond[k], scale);
}
bool sd_set_vae_conv2d_scale(sdm_ctx_t* ctx, float scale) {
    if (!ctx || !(scale > 0.f) || !std::isfinite(scale)) {
        set_error("sd_set_vae_conv2d_scale: the scale must be a finite positive number");
        return false;
    }
    ctx->vae_conv2d_scale = scale;
    ctx->vae.set_conv2d_scale(scale);
    if (ctx->vae_enc_ready) ctx->vae_enc.set_conv2d_scale(scale);
    return true;
}
void sd_set_pair_exchange(sdm_ctx_t* ctx, sd_pair_exchange_fn fn, void* user, int branch) {
    ctx->pair_fn     = fn;
    ctx->pair_user   = user;
    ctx->pair_branch = branch != 0;
}
int sd_get_flux_sigmas(int steps, int image_seq_len, float* out) {
    FluxFlowDenoiser d;
    std::vector<float> s = d.get_sigmas(steps, image_seq_len);
    memcpy(out, s.data(), s.size() * sizeof(float));
    return (int)s.size();
}
int sd_gen_flux_pe(int h, int w, int patch_size, int context_len, const int* axes_dim, int n_axes, float theta, float* out) {
    std::vector<float> pe = gen_flux_pe(h, w, patch_size, context_len, std::vector<int>(axes_dim, axes_dim + n_axes), theta);
    memcpy(out, pe.data(), pe.size() * sizeof(float));
    return (int)pe.size();
}
int sd_get_flow_sigmas(int steps, float shift, float* out) {
    DiscreteFlowDenoiser d;
    d.shift              = shift;
    std::vector<float> s = d.get_sigmas(steps);
    memcpy(out, s.data(), s.size() * sizeof(float));
    return (int)s.size();
}
float sd_sigma_to_t(float sigma) {
    static CompVisDenoiser d;
    return d.sigma_to_t(sigma);
}
void sd_get_stats(sdm_ctx_t* ctx, sd_stats_t* out) {
    ctx->stats.host_build_ms  = ctx->unet_runner.build_ms;
    ctx->stats.host_alloc_ms  = ctx->unet_runner.alloc_ms;
    ctx->stats.host_submit_ms = ctx->unet_runner.submit_ms;
    ctx->stats.graph_cache_hits = ctx->unet_runner.cache_hits;
    *out                      = ctx->stats;
}

}  // extern "C"
