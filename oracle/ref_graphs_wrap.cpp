// TEST INFRASTRUCTURE (oracle/): the REFERENCE's own graph builders and runner, compiled from where they lie under /root/reference by oracle/Makefile
// (target _ref/libref_graphs.so) and linked against THIS repository's ggml front-end (stable-diffusion.cpp_amd/lib/libsdcpp-host.so = csrc/ggml):
//
//   src/core/ggml_extend.hpp            ggml_ext_* op wrappers, GGMLBlock / Linear / Conv2d / norms, GGMLRunner (graph lifecycle: compute ctx, built-in
//                                       leaves, final-result naming, gallocr placement, input upload, graph_compute through the eval-callback loop, read-back)
//   src/core/ggml_extend_backend.cpp    sd_backend_graph_compute_with_eval_callback / sd_ggml_graph_view (the sub-graph-view contract), backend helpers
//   src/core/ggml_graph_cut.cpp, layer_split_partition.cpp, util.cpp     (what GGMLRunner links)
//   src/model/common/block.hpp, rope.hpp; src/model/diffusion/unet.hpp (UnetModelBlock, UNetModelRunner), mmdit.hpp (MMDiT, MMDiTRunner),
//   flux.hpp (Flux, FluxRunner), dit.hpp; src/model/vae/auto_encoder_kl.hpp (AutoEncoderKLModel, AutoEncoderKL), vae.hpp
//
// Nothing of the reference is copied into this repository; nothing here is loaded by the product.  What the library is for (tests/test_ref_graphs.py,
// tests/test_gpu_ref_graphs.py):
//   (a) TOPOLOGY ORACLE — the graph the reference emits for a model / shape, described by the host's sdm_graph_describe, must equal node for node (op,
//       type, ne, nb, op_params, flags, source indices, view offsets, parameter names) the graph csrc/host/models.hpp emits for the same configuration;
//   (b) the reference's runner drives a registered ggml backend (the MI355X plug-in or the CPU oracle) through ITS OWN GGMLRunner::compute: its gallocr
//       use, its input uploads, its eval-callback slicing — the offline half of "real-ggml integration" (SURVEY.md section 8 row f1).
// The only piece written here instead of taken from the reference is weight residency: the reference streams weights through a ModelManager
// (src/model_manager.cpp: model files, LoRA, offload); ResidentWeights below keeps every parameter in one backend buffer the test fills.
//
// ggml API the reference headers mention but these paths never call is declared in oracle/ggml_api/ggml-extra-decls.h; the definitions at the end of this
// file abort with the function's name (reaching one cannot silently compute something).
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "model/diffusion/unet.hpp"
#include "model/vae/auto_encoder_kl.hpp"
#include "model/vae/tae.hpp"
#include "model/diffusion/dit.hpp"
#include "model/diffusion/mmdit.hpp"
#include "model/diffusion/flux.hpp"

// the host library's describer (include/sd-mi355x.h) — one serialiser for both sides of the comparison
extern "C" size_t sdm_graph_describe(struct ggml_cgraph* gf, char* buf, size_t cap);

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

String2TensorStorage g_storage;  // the "model file" table the reference's constructors read types and shapes from

struct ResidentWeights : public RunnerWeightManager {
    bool assign_compute_backend(const std::vector<ggml_tensor*>&, ggml_backend_t) override { return true; }
    bool prepare_params(const std::vector<ggml_tensor*>& tensors) override {
        for (ggml_tensor* t : tensors)
            if (t != nullptr && t->buffer == nullptr && (t->view_src == nullptr || t->view_src->buffer == nullptr)) return false;  // refg_alloc_params was not called
        return true;
    }
    void release_compute_backend_params(const std::vector<ggml_tensor*>&) override {}
    void release_params_backend_params(const std::vector<ggml_tensor*>&) override {}
};

std::map<std::string, std::string> parse_overrides(const char* s) {
    std::map<std::string, std::string> kv;
    if (!s) return kv;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, ';')) {
        const size_t eq = item.find('=');
        if (eq != std::string::npos) kv[item.substr(0, eq)] = item.substr(eq + 1);
    }
    return kv;
}
std::vector<int> int_list(const std::string& v) {
    std::vector<int> out;
    std::stringstream ss(v);
    std::string item;
    while (std::getline(ss, item, ',')) out.push_back(atoi(item.c_str()));
    return out;
}

struct Handle {
    int family = 0;
    ggml_backend_t backend = nullptr;
    std::shared_ptr<ResidentWeights> weights = std::make_shared<ResidentWeights>();
    std::vector<std::pair<std::string, ggml_tensor*>> params;
    ggml_backend_buffer_t params_buffer = nullptr;
    std::string described;
    virtual ~Handle() {
        if (params_buffer) ggml_backend_buffer_free(params_buffer);
    }
    virtual ggml_context* params_context() = 0;
    void list_params(std::map<std::string, ggml_tensor*>& m) {
        for (auto& kv : m) params.push_back(kv);
    }
};

// Each runner is the reference's own class.  The constructor detects the configuration from the storage table exactly as the reference does; overrides
// (the test models' reduced widths set fields no weight shape reveals: head counts, transformer depths, RoPE axes) rebuild the block with the edited
// configuration in a fresh parameter context — the reference's own build_graph / compute run unchanged either way.
struct UNetH : public Handle, public UNetModelRunner {
    UNetH(ggml_backend_t be, const std::string& prefix, SDVersion v, const std::map<std::string, std::string>& ov, std::shared_ptr<ResidentWeights> w)
        : UNetModelRunner(be, g_storage, prefix, v, w) {
        weights = w;
        if (!ov.empty()) {
            UNetConfig c = config;
            for (auto& [k, val] : ov) {
                if (k == "num_heads") c.num_heads = atoi(val.c_str());
                else if (k == "num_head_channels") c.num_head_channels = atoi(val.c_str());
                else if (k == "transformer_depth") c.transformer_depth = int_list(val);
                else if (k == "channel_mult") c.channel_mult = int_list(val);
                else if (k == "attention_resolutions") c.attention_resolutions = int_list(val);
                else if (k == "num_res_blocks") c.num_res_blocks = atoi(val.c_str());
                else if (k == "context_dim") c.context_dim = atoi(val.c_str());
                else if (k == "use_linear_projection") c.use_linear_projection = atoi(val.c_str()) != 0;
                else GGML_ABORT("unknown UNet override %s", k.c_str());
            }
            free_params_ctx();
            alloc_params_ctx();
            config = c;
            unet   = UnetModelBlock(c);
            unet.init(params_ctx, g_storage, prefix);
        }
        std::map<std::string, ggml_tensor*> m;
        UNetModelRunner::get_param_tensors(m, prefix);
        list_params(m);
    }
    ggml_context* params_context() override { return params_ctx; }
    using GGMLRunner::prepare_compute_graph;
    using GGMLRunner::free_compute_ctx;
};

struct MMDiTH : public Handle, public MMDiTRunner {
    MMDiTH(ggml_backend_t be, const std::string& prefix, const std::map<std::string, std::string>& ov, std::shared_ptr<ResidentWeights> w) : MMDiTRunner(be, g_storage, prefix, w) {
        weights = w;
        if (!ov.empty()) {
            MMDiTConfig c = config;
            for (auto& [k, val] : ov) {
                if (k == "depth") c.depth = atoi(val.c_str());
                else if (k == "d_self") c.d_self = atoi(val.c_str());
                else GGML_ABORT("unknown MMDiT override %s", k.c_str());
            }
            free_params_ctx();
            alloc_params_ctx();
            config = c;
            mmdit  = MMDiT(c);
            mmdit.init(params_ctx, g_storage, prefix);
        }
        std::map<std::string, ggml_tensor*> m;
        MMDiTRunner::get_param_tensors(m, prefix);
        list_params(m);
    }
    ggml_context* params_context() override { return params_ctx; }
    using GGMLRunner::prepare_compute_graph;
    using GGMLRunner::free_compute_ctx;
};

struct FluxH : public Handle, public Flux::FluxRunner {
    FluxH(ggml_backend_t be, const std::string& prefix, const std::map<std::string, std::string>& ov, std::shared_ptr<ResidentWeights> w)
        : Flux::FluxRunner(be, g_storage, prefix, VERSION_FLUX, w) {
        weights = w;
        if (!ov.empty()) {
            Flux::FluxConfig c = config;
            for (auto& [k, val] : ov) {
                if (k == "vec_in_dim") c.vec_in_dim = atoi(val.c_str());
                else if (k == "axes_dim") c.axes_dim = int_list(val);
                else if (k == "num_heads") c.num_heads = atoi(val.c_str());
                else GGML_ABORT("unknown FLUX override %s", k.c_str());
            }
            c.axes_dim_sum = 0;
            for (int a : c.axes_dim) c.axes_dim_sum += a;
            free_params_ctx();
            alloc_params_ctx();
            config = c;
            flux   = Flux::Flux(c);
            flux.init(params_ctx, g_storage, prefix);
        }
        std::map<std::string, ggml_tensor*> m;
        Flux::FluxRunner::get_param_tensors(m, prefix);
        list_params(m);
    }
    ggml_context* params_context() override { return params_ctx; }
    using GGMLRunner::prepare_compute_graph;
    using GGMLRunner::free_compute_ctx;
};

struct VaeH : public Handle, public AutoEncoderKL {
    VaeH(ggml_backend_t be, const std::string& prefix, SDVersion v, std::shared_ptr<ResidentWeights> w, bool with_encoder = false)
        : AutoEncoderKL(be, g_storage, prefix, /*decode_only*/ !with_encoder, false, v, w) {
        weights = w;
        std::map<std::string, ggml_tensor*> m;
        AutoEncoderKL::get_param_tensors(m);
        list_params(m);
    }
    ggml_context* params_context() override { return params_ctx; }
    using GGMLRunner::prepare_compute_graph;
    using GGMLRunner::free_compute_ctx;
};

// TAESD (round 6): the reference makes it with the lookup prefix "decoder.layers" and names its parameters under "tae" (stable-diffusion.cpp:1417-1422, tae.hpp:744, 753)
struct TaeH : public Handle, public TinyImageAutoEncoder {
    TaeH(ggml_backend_t be, SDVersion v, std::shared_ptr<ResidentWeights> w) : TinyImageAutoEncoder(be, g_storage, "decoder.layers", /*decoder_only*/ true, v, w) {
        weights = w;
        std::map<std::string, ggml_tensor*> m;
        TinyImageAutoEncoder::get_param_tensors(m);
        list_params(m);
    }
    ggml_context* params_context() override { return params_ctx; }
    using GGMLRunner::prepare_compute_graph;
    using GGMLRunner::free_compute_ctx;
};

sd::Tensor<float> tensor_of(const float* data, const int64_t* ne, int n_dims) {
    if (!data || n_dims <= 0) return {};
    std::vector<int64_t> shape(ne, ne + n_dims);
    int64_t n = 1;
    for (int64_t d : shape) n *= d;
    return sd::Tensor<float>(shape, std::vector<float>(data, data + n));
}

SDVersion version_of(const char* v) {
    const std::string s = v ? v : "sd1";
    if (s == "sd1") return VERSION_SD1;
    if (s == "sdxl") return VERSION_SDXL;
    if (s == "sd3") return VERSION_SD3;
    if (s == "flux") return VERSION_FLUX;
    GGML_ABORT("unknown version %s", s.c_str());
}

}  // namespace

// ---- the storage table ("model file" index): name -> type, shape --------------------------------------------------------------------------------------
REF_API void refg_storage_clear(void) { g_storage = String2TensorStorage(); }
REF_API void refg_storage_add(const char* name, int type, int n_dims, const int64_t* ne) {
    TensorStorage ts(name, (ggml_type)type, ne, n_dims, 0, 0);
    g_storage[name] = ts;
}

// family: 0 UNet, 1 KL-VAE decoder, 2 MMDiT, 3 FLUX, 4 TAESD decoder, 5 KL-VAE with its encoder (encode graph).  overrides: "key=value;key=a,b,c" applied on top of the detected configuration ("" / NULL: none)
REF_API void* refg_new(int family, const char* version, void* backend, const char* prefix, int flash_attn, const char* overrides) {
    auto w  = std::make_shared<ResidentWeights>();
    auto ov = parse_overrides(overrides);
    auto be = (ggml_backend_t)backend;
    Handle* h = nullptr;
    GGMLRunner* r = nullptr;
    switch (family) {
        case 0: {
            auto* p = new UNetH(be, prefix, version_of(version), ov, w);
            h = p, r = p;
            break;
        }
        case 1: {
            auto* p = new VaeH(be, prefix, version_of(version), w);
            h = p, r = p;
            break;
        }
        case 2: {
            auto* p = new MMDiTH(be, prefix, ov, w);
            h = p, r = p;
            break;
        }
        case 3: {
            auto* p = new FluxH(be, prefix, ov, w);
            h = p, r = p;
            break;
        }
        case 4: {
            auto* p = new TaeH(be, version_of(version), w);
            h = p, r = p;
            break;
        }
        case 5: {  // the whole autoencoder (decode_only = false): the ENCODE graph is what refg_run builds
            auto* p = new VaeH(be, prefix, version_of(version), w, true);
            h = p, r = p;
            break;
        }
        default: return nullptr;
    }
    h->family  = family;
    h->backend = be;
    r->set_flash_attention_enabled(flash_attn != 0);
    return h;
}
REF_API void refg_free(void* hp) { delete (Handle*)hp; }
REF_API int64_t refg_param_count(void* hp) { return (int64_t)((Handle*)hp)->params.size(); }
REF_API const char* refg_param_name(void* hp, int64_t i) { return ((Handle*)hp)->params[(size_t)i].first.c_str(); }
REF_API void* refg_param_tensor(void* hp, int64_t i) { return ((Handle*)hp)->params[(size_t)i].second; }
// every parameter into ONE buffer of the runner's backend, flagged WEIGHTS (what the reference's ModelManager does per tensor group)
REF_API int refg_alloc_params(void* hp) {
    Handle* h = (Handle*)hp;
    if (h->params_buffer) return 1;
    h->params_buffer = ggml_backend_alloc_ctx_tensors(h->params_context(), h->backend);
    if (!h->params_buffer) return 0;
    ggml_backend_buffer_set_usage(h->params_buffer, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
    return 1;
}
REF_API void refg_set_conv2d_scale(void* hp, float scale) {
    Handle* h = (Handle*)hp;
    if (h->family == 1 || h->family == 5) static_cast<VaeH*>(h)->set_conv2d_scale(scale);
}

// Inputs as the reference's sd::Tensor<float> (shape = ggml ne order): x [W,H,C,N], timesteps [N], context [dim, L, Nc], y [dim, Ny], guidance [N] (FLUX).
// describe_only: build the graph through the reference's prepare_compute_graph (compute ctx, built-in leaves, final-result name) and return its
// description (refg_description); otherwise run the reference's compute() and copy the result to `out` (returns its element count, -1 on failure).
REF_API int64_t refg_run(void* hp, int describe_only, const float* x, const int64_t* x_ne, const float* t, int64_t n_t, const float* ctx, const int64_t* ctx_ne, const float* y,
                         const int64_t* y_ne, const float* guidance, int64_t n_g, float* out, int64_t out_cap) {
    Handle* h  = (Handle*)hp;
    auto X     = tensor_of(x, x_ne, 4);
    auto T     = tensor_of(t, &n_t, 1);
    auto C     = ctx ? tensor_of(ctx, ctx_ne, 3) : sd::Tensor<float>();
    auto Y     = y ? tensor_of(y, y_ne, 2) : sd::Tensor<float>();
    auto G     = guidance ? tensor_of(guidance, &n_g, 1) : sd::Tensor<float>();
    sd::Tensor<float> result;
    std::function<ggml_cgraph*()> get_graph;
    GGMLRunner* r = nullptr;
    switch (h->family) {
        case 0: {
            auto* p   = static_cast<UNetH*>(h);
            r         = p;
            get_graph = [&, p]() { return p->build_graph(X, T, C, {}, Y); };
            if (!describe_only) result = p->UNetModelRunner::compute(1, X, T, C, {}, Y);
            break;
        }
        case 1: {
            auto* p   = static_cast<VaeH*>(h);
            r         = p;
            get_graph = [&, p]() { return p->build_graph(X, true); };
            if (!describe_only) result = p->_compute(1, X, true);
            break;
        }
        case 2: {
            auto* p   = static_cast<MMDiTH*>(h);
            r         = p;
            get_graph = [&, p]() { return p->build_graph(X, T, C, Y); };
            if (!describe_only) result = p->MMDiTRunner::compute(1, X, T, C, Y);
            break;
        }
        case 3: {
            auto* p   = static_cast<FluxH*>(h);
            r         = p;
            get_graph = [&, p]() { return p->build_graph(X, T, C, {}, Y, G); };
            if (!describe_only) result = p->Flux::FluxRunner::compute(1, X, T, C, {}, Y, G);
            break;
        }
        case 4: {
            auto* p   = static_cast<TaeH*>(h);
            r         = p;
            get_graph = [&, p]() { return p->build_graph(X, true); };
            if (!describe_only) result = p->_compute(1, X, true);
            break;
        }
        case 5: {
            auto* p   = static_cast<VaeH*>(h);
            r         = p;
            get_graph = [&, p]() { return p->build_graph(X, false); };
            if (!describe_only) result = p->_compute(1, X, false);
            break;
        }
    }
    if (describe_only) {
        ggml_cgraph* gf = nullptr;
        bool ok         = false;
        switch (h->family) {
            case 0: ok = static_cast<UNetH*>(h)->prepare_compute_graph(get_graph, &gf); break;
            case 1: ok = static_cast<VaeH*>(h)->prepare_compute_graph(get_graph, &gf); break;
            case 2: ok = static_cast<MMDiTH*>(h)->prepare_compute_graph(get_graph, &gf); break;
            case 3: ok = static_cast<FluxH*>(h)->prepare_compute_graph(get_graph, &gf); break;
            case 4: ok = static_cast<TaeH*>(h)->prepare_compute_graph(get_graph, &gf); break;
            case 5: ok = static_cast<VaeH*>(h)->prepare_compute_graph(get_graph, &gf); break;
        }
        if (!ok || !gf) return -1;
        const size_t need = sdm_graph_describe(gf, nullptr, 0);
        h->described.assign(need, '\0');
        sdm_graph_describe(gf, h->described.data(), need);
        h->described.resize(need - 1);
        const int64_t n_nodes = ggml_graph_n_nodes(gf);
        switch (h->family) {
            case 0: static_cast<UNetH*>(h)->free_compute_ctx(); break;
            case 1: static_cast<VaeH*>(h)->free_compute_ctx(); break;
            case 2: static_cast<MMDiTH*>(h)->free_compute_ctx(); break;
            case 3: static_cast<FluxH*>(h)->free_compute_ctx(); break;
            case 4: static_cast<TaeH*>(h)->free_compute_ctx(); break;
            case 5: static_cast<VaeH*>(h)->free_compute_ctx(); break;
        }
        return n_nodes;
    }
    (void)r;
    if (result.empty()) return -1;
    const int64_t n = result.numel();
    if (out && n <= out_cap) memcpy(out, result.data(), (size_t)n * sizeof(float));
    return n;
}
REF_API const char* refg_description(void* hp) { return ((Handle*)hp)->described.c_str(); }

// the reference's own node-by-node hook (src/core/util.cpp:638-668 -> sd_backend_graph_compute_with_eval_callback, src/core/ggml_extend_backend.cpp:466-509)
REF_API void refg_set_eval_callback(bool (*cb)(struct ggml_tensor*, bool, void*), void* user) { sd_set_backend_eval_callback(cb, user); }

// ---- ggml API the reference headers mention and these paths never reach (oracle/ggml_api/ggml-extra-decls.h).  ctypes loads libraries with RTLD_NOW, so
// the symbols must resolve: each one aborts with its name — a graph that reached one of them could not silently compute something.
#define REFG_UNREACHED(ret, name, args) \
    extern "C" ret name args { GGML_ABORT(#name ": not implemented by this repository's ggml front-end (not on the UNet / MMDiT / FLUX / VAE path)"); }
REFG_UNREACHED(void, ggml_log_set, (ggml_log_callback, void*))
REFG_UNREACHED(ggml_tensor*, ggml_pad_ext_circular, (ggml_context*, ggml_tensor*, int, int, int, int, int, int, int, int))
REFG_UNREACHED(ggml_tensor*, ggml_roll, (ggml_context*, ggml_tensor*, int, int, int, int))
REFG_UNREACHED(ggml_tensor*, ggml_interpolate, (ggml_context*, ggml_tensor*, int64_t, int64_t, int64_t, int64_t, uint32_t))
REFG_UNREACHED(ggml_tensor*, ggml_arange, (ggml_context*, float, float, float))
REFG_UNREACHED(ggml_tensor*, ggml_im2col_3d, (ggml_context*, ggml_tensor*, ggml_tensor*, int64_t, int, int, int, int, int, int, int, int, int, enum ggml_type))
REFG_UNREACHED(ggml_tensor*, ggml_conv_3d, (ggml_context*, ggml_tensor*, ggml_tensor*, int64_t, int, int, int, int, int, int, int, int, int))
REFG_UNREACHED(ggml_tensor*, ggml_conv_3d_direct, (ggml_context*, ggml_tensor*, ggml_tensor*, int, int, int, int, int, int, int, int, int, int, int, int))
REFG_UNREACHED(ggml_tensor*, ggml_conv_2d_dw, (ggml_context*, ggml_tensor*, ggml_tensor*, int, int, int, int, int, int))
REFG_UNREACHED(ggml_tensor*, ggml_conv_2d_dw_direct, (ggml_context*, ggml_tensor*, ggml_tensor*, int, int, int, int, int, int))
REFG_UNREACHED(ggml_tensor*, ggml_mul_mat_i8_tensorwise, (ggml_context*, ggml_tensor*, ggml_tensor*, ggml_tensor*, ggml_tensor*, int))
REFG_UNREACHED(ggml_tensor*, ggml_l2_norm, (ggml_context*, ggml_tensor*, float))
REFG_UNREACHED(ggml_tensor*, ggml_quantize_i8_convrot, (ggml_context*, ggml_tensor*, int))
REFG_UNREACHED(ggml_backend_sched_t, ggml_backend_sched_new, (ggml_backend_t*, ggml_backend_buffer_type_t*, int, size_t, bool, bool))
REFG_UNREACHED(void, ggml_backend_sched_free, (ggml_backend_sched_t))
REFG_UNREACHED(void, ggml_backend_sched_reset, (ggml_backend_sched_t))
REFG_UNREACHED(void, ggml_backend_sched_synchronize, (ggml_backend_sched_t))
REFG_UNREACHED(bool, ggml_backend_sched_alloc_graph, (ggml_backend_sched_t, ggml_cgraph*))
REFG_UNREACHED(enum ggml_status, ggml_backend_sched_graph_compute, (ggml_backend_sched_t, ggml_cgraph*))
REFG_UNREACHED(void, ggml_backend_sched_set_tensor_backend, (ggml_backend_sched_t, ggml_tensor*, ggml_backend_t))
REFG_UNREACHED(void, ggml_gallocr_reserve_n_size, (ggml_gallocr_t, ggml_cgraph*, const int*, const int*, size_t*))
