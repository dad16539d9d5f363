// nn.hpp — host-side graph builders for the layers on the hot path.  Each builder emits the SAME ggml
// node sequence as the reference wrapper it mirrors (cited per function), because the node sequence
// *is* the contract the backend sees behind ggml_backend_graph_compute.
//
// Design differs from the reference's GGMLBlock class tree (src/core/ggml_extend.hpp:3280-3400): here a
// layer is a plain struct that registers its parameter tensors by name in a ParamStore and exposes a
// forward() that appends nodes to the current graph context.  Tensor names follow the reference's
// checkpoint naming (e.g. "input_blocks.1.0.in_layers.2.weight") so real weights can be bound later
// (SURVEY.md §8 f2).
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ggml.h"

namespace sdmi {

enum class InitKind { WEIGHT, BIAS, NORM_SCALE, ZERO };

struct ParamSpec {
    std::string name;
    ggml_tensor* tensor;
    InitKind kind;
    int64_t fan_in;
};

// Owns the (no_alloc) params context; tensors are placed in ONE backend buffer with usage WEIGHTS
// (model_manager.cpp:735-750) and filled by synthetic init or by the caller.
struct ParamStore {
    ggml_context* ctx = nullptr;
    std::vector<ParamSpec> specs;
    std::map<std::string, ggml_tensor*> by_name;
    ggml_type linear_type = GGML_TYPE_F16;  // sdm_ctx_params_t.wtype analogue for Linear weights

    ParamStore() {
        ggml_init_params p{0, nullptr, true};
        ctx = ggml_init(p);
    }
    ~ParamStore() { ggml_free(ctx); }
    ParamStore(const ParamStore&) = delete;

    ggml_tensor* add(const std::string& name, ggml_type type, std::vector<int64_t> ne, InitKind kind, int64_t fan_in) {
        ggml_tensor* t = ggml_new_tensor(ctx, type, (int)ne.size(), ne.data());
        ggml_set_name(t, name.c_str());
        specs.push_back({name, t, kind, fan_in});
        by_name[name] = t;
        return t;
    }
};

// Per-graph build context (the reference's GGMLRunnerContext, ggml_extend.hpp:1724-1760)
struct GraphCtx {
    ggml_context* ctx      = nullptr;
    ggml_backend_t backend = nullptr;
    bool flash_attn        = false;  // sdm_ctx_params_t.diffusion_flash_attn
    bool conv_direct       = false;  // sdm_ctx_params_t.diffusion_conv_direct
    const std::vector<int>* skip_layers = nullptr;  // skip-layer guidance: joint blocks the MMDiT forward leaves out (mmdit.hpp:854-866)
};

// ---- ggml_ext_* wrappers (node-for-node) --------------------------------------------------------
inline bool is_padded_1d(const ggml_tensor* x) {  // ggml_extend.hpp:959-963
    return x->nb[0] == ggml_type_size(x->type) && x->nb[2] == x->nb[1] * x->ne[1] && x->nb[3] == x->nb[2] * x->ne[2];
}
inline ggml_tensor* ext_scale(ggml_context* c, ggml_tensor* x, float f, bool inplace = false) {  // :965-978
    if (!is_padded_1d(x)) x = ggml_cont(c, x);
    return inplace ? ggml_scale_inplace(c, x, f) : ggml_scale(c, x, f);
}
inline ggml_tensor* ext_cont(ggml_context* c, ggml_tensor* x) { return ggml_is_contiguous(x) ? x : ggml_cont(c, x); }  // :572-578
inline ggml_tensor* ext_gelu(ggml_context* c, ggml_tensor* x, bool inplace) {                                       // :980-992
    if (!ggml_is_contiguous(x)) x = ggml_cont(c, x);
    return inplace ? ggml_gelu_inplace(c, x) : ggml_gelu(c, x);
}
inline std::vector<ggml_tensor*> ext_chunk(ggml_context* c, ggml_tensor* x, int num, int64_t dim, bool cont) {  // :641-666
    std::vector<ggml_tensor*> out;
    const int64_t chunk = x->ne[dim] / num;
    const size_t stride = chunk * x->nb[dim];
    int64_t ne[4]       = {x->ne[0], x->ne[1], x->ne[2], x->ne[3]};
    ne[dim]             = chunk;
    for (int i = 0; i < num; ++i) {
        ggml_tensor* v = ggml_view_4d(c, x, ne[0], ne[1], ne[2], ne[3], x->nb[1], x->nb[2], x->nb[3], stride * i);
        out.push_back(cont ? ggml_cont(c, v) : v);
    }
    return out;
}

// ggml_ext_linear — ggml_extend.hpp:1008-1040
inline ggml_tensor* ext_linear(ggml_context* c, ggml_tensor* x, ggml_tensor* w, ggml_tensor* b, bool force_prec_f32 = false, float scale = 1.f) {
    if (scale != 1.f) x = ext_scale(c, x, scale);
    if (x->ne[2] * x->ne[3] > 1024) {
        const int64_t ne2 = x->ne[2], ne3 = x->ne[3];
        x = ggml_reshape_2d(c, x, x->ne[0], x->ne[1] * x->ne[2] * x->ne[3]);
        x = ggml_mul_mat(c, w, x);
        if (force_prec_f32) ggml_mul_mat_set_prec(x, GGML_PREC_F32);
        x = ggml_reshape_4d(c, x, x->ne[0], x->ne[1] / ne2 / ne3, ne2, ne3);
    } else {
        x = ggml_mul_mat(c, w, x);
        if (force_prec_f32) ggml_mul_mat_set_prec(x, GGML_PREC_F32);
    }
    if (scale != 1.f) x = ext_scale(c, x, 1.f / scale);
    if (b != nullptr) x = ggml_add_inplace(c, x, b);
    return x;
}

// ggml_ext_conv_2d — ggml_extend.hpp:1131-1171 (non-circular)
inline ggml_tensor* ext_conv_2d(ggml_context* c, ggml_tensor* x, ggml_tensor* w, ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1, bool direct, float scale = 1.f) {
    if (scale != 1.f) x = ext_scale(c, x, scale);
    if (w->ne[2] != x->ne[2] && ggml_n_dims(w) == 2) w = ggml_reshape_4d(c, w, 1, 1, w->ne[0], w->ne[1]);
    x = direct ? ggml_conv_2d_direct(c, w, x, s0, s1, p0, p1, d0, d1) : ggml_conv_2d(c, w, x, s0, s1, p0, p1, d0, d1);
    if (scale != 1.f) x = ext_scale(c, x, 1.f / scale);
    if (b != nullptr) {
        b = ggml_reshape_4d(c, b, 1, 1, b->ne[0], 1);
        x = ggml_add_inplace(c, x, b);
    }
    return x;
}

// ggml_ext_layer_norm — ggml_extend.hpp:1487-1500
inline ggml_tensor* ext_layer_norm(ggml_context* c, ggml_tensor* x, ggml_tensor* w, ggml_tensor* b, float eps) {
    x = ggml_norm(c, x, eps);
    if (w != nullptr) {
        x = ggml_mul_inplace(c, x, w);
        if (b != nullptr) x = ggml_add_inplace(c, x, b);
    }
    return x;
}

// ggml_ext_group_norm — ggml_extend.hpp:1502-1520 (eps fixed at 1e-6)
inline ggml_tensor* ext_group_norm(ggml_context* c, ggml_tensor* x, ggml_tensor* w, ggml_tensor* b, int num_groups = 32) {
    if (ggml_n_dims(x) >= 3 && w != nullptr && b != nullptr) {
        w = ggml_reshape_4d(c, w, 1, 1, w->ne[0], 1);
        b = ggml_reshape_4d(c, b, 1, 1, b->ne[0], 1);
    }
    x = ggml_group_norm(c, x, num_groups, 1e-6f);
    if (w != nullptr && b != nullptr) {
        x = ggml_mul_inplace(c, x, w);
        x = ggml_add_inplace(c, x, b);
    }
    return x;
}

// ggml_ext_attention_ext — ggml_extend.hpp:1349-1485 (kv_scale = 1)
// skip_reshape (ggml_extend.hpp:1383-1390): q, k arrive as [d_head, L, n_head*N] and v as [d_head, n_head, L, N] (Rope::attention)
// mask (text encoders): f32 [L_k, L_q or 1, heads or 1, 1] added to the scaled scores; the flash path materialises the query dimension
// and casts it to f16 (:1411-1425), the manual path adds it in place (:1466-1468)
inline ggml_tensor* ext_attention(GraphCtx& g, ggml_tensor* q, ggml_tensor* k, ggml_tensor* v, int64_t n_head, bool skip_reshape = false, ggml_tensor* mask = nullptr) {
    ggml_context* c = g.ctx;
    int64_t L_q = q->ne[1], L_k = k->ne[1], C = q->ne[0], N = q->ne[2];
    int64_t d_head = C / n_head, n_kv_head = k->ne[0] / d_head;
    if (!skip_reshape) {
        q = ggml_reshape_4d(c, q, d_head, n_head, L_q, N);
        q = ext_cont(c, ggml_permute(c, q, 0, 2, 1, 3));
        q = ggml_reshape_3d(c, q, d_head, L_q, n_head * N);
        k = ggml_reshape_4d(c, k, d_head, n_kv_head, L_k, N);
        k = ext_cont(c, ggml_permute(c, k, 0, 2, 1, 3));
        k = ggml_reshape_3d(c, k, d_head, L_k, n_kv_head * N);
        v = ggml_reshape_4d(c, v, d_head, n_kv_head, L_k, N);
    } else {
        d_head    = v->ne[0];
        N         = v->ne[3];
        n_kv_head = k->ne[2] / N;
        C         = d_head * n_head;
    }

    const float scale = 1.0f / sqrtf((float)d_head);
    ggml_tensor* kqv  = nullptr;
    if (g.flash_attn && (mask == nullptr || mask->ne[3] == 1)) {
        ggml_tensor* k_in = ggml_cast(c, k, GGML_TYPE_F16);
        ggml_tensor* v_in = ext_cont(c, ggml_permute(c, v, 0, 2, 1, 3));
        v_in              = ggml_reshape_3d(c, v_in, d_head, L_k, n_kv_head * N);
        v_in              = ggml_cast(c, v_in, GGML_TYPE_F16);
        ggml_tensor* m_in = mask;
        if (m_in != nullptr) {
            if (m_in->ne[1] != L_q) m_in = ggml_repeat(c, m_in, ggml_new_tensor_4d(c, m_in->type, m_in->ne[0], L_q, m_in->ne[2], m_in->ne[3]));
            m_in = ggml_cast(c, m_in, GGML_TYPE_F16);
        }
        ggml_tensor* out  = ggml_flash_attn_ext(c, q, k_in, v_in, m_in, scale, 0, 0);
        if (g.backend == nullptr || ggml_backend_supports_op(g.backend, out)) {
            ggml_flash_attn_ext_set_prec(out, GGML_PREC_F32);
            kqv = ggml_view_4d(c, out, d_head, n_head, L_q, N, out->nb[1], out->nb[2], out->nb[1] * n_head, 0);
        }
    }
    if (kqv == nullptr) {
        v               = ext_cont(c, ggml_permute(c, v, 1, 2, 0, 3));
        v               = ggml_reshape_3d(c, v, L_k, d_head, n_kv_head * N);
        ggml_tensor* kq = ggml_mul_mat(c, k, q);
        ggml_mul_mat_set_prec(kq, GGML_PREC_F32);
        kq  = ggml_scale_inplace(c, kq, scale);
        if (mask) kq = ggml_add_inplace(c, kq, mask);
        kq  = ggml_soft_max_inplace(c, kq);
        kqv = ggml_mul_mat(c, v, kq);
        kqv = ggml_reshape_4d(c, kqv, d_head, L_q, n_head, N);
        kqv = ggml_permute(c, kqv, 0, 2, 1, 3);
    }
    kqv = ext_cont(c, kqv);
    kqv = ggml_reshape_3d(c, kqv, d_head * n_head, L_q, N);
    return kqv;
}

// ---- layers ---------------------------------------------------------------------------------------
// Linear — ggml_extend.hpp:3403-3536 (weight takes wtype iff in % blck == 0; bias always F32)
struct Linear {
    ggml_tensor *w = nullptr, *b = nullptr;
    bool force_prec_f32 = false;
    float scale         = 1.f;
    void init(ParamStore& ps, const std::string& prefix, int64_t in, int64_t out, bool bias = true, bool never_quant = false, bool force_f32 = false) {
        ggml_type t = ps.linear_type;
        if (never_quant && ggml_is_quantized(t)) t = GGML_TYPE_F16;  // model_loader.cpp:1517-1539
        if (in % ggml_blck_size(t) != 0 || force_f32) t = GGML_TYPE_F32;  // ggml_extend.hpp:3422-3425
        w = ps.add(prefix + "weight", t, {in, out}, InitKind::WEIGHT, in);
        if (bias) b = ps.add(prefix + "bias", GGML_TYPE_F32, {out}, InitKind::BIAS, in);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ext_linear(g.ctx, x, w, b, force_prec_f32, scale); }
};

// Conv2d — ggml_extend.hpp:3588-3669 (weight always F16, bias F32)
struct Conv2d {
    ggml_tensor *w = nullptr, *b = nullptr;
    int s = 1, p = 0;
    float scale = 1.f;
    void init(ParamStore& ps, const std::string& prefix, int64_t in, int64_t out, int k, int stride = 1, int pad = 0, bool bias = true) {
        s = stride;
        p = pad;
        w = ps.add(prefix + "weight", GGML_TYPE_F16, {k, k, in, out}, InitKind::WEIGHT, in * k * k);
        if (bias) b = ps.add(prefix + "bias", GGML_TYPE_F32, {out}, InitKind::BIAS, in * k * k);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ext_conv_2d(g.ctx, x, w, b, s, s, p, p, 1, 1, g.conv_direct, scale); }
};

// GroupNorm32 — ggml_extend.hpp:3946-3994
struct GroupNorm32 {
    ggml_tensor *w = nullptr, *b = nullptr;
    void init(ParamStore& ps, const std::string& prefix, int64_t ch) {
        w = ps.add(prefix + "weight", GGML_TYPE_F32, {ch}, InitKind::NORM_SCALE, ch);
        b = ps.add(prefix + "bias", GGML_TYPE_F32, {ch}, InitKind::BIAS, ch);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ext_group_norm(g.ctx, x, w, b, 32); }
};

// LayerNorm — ggml_extend.hpp:3897-3944 (eps 1e-5)
struct LayerNorm {
    ggml_tensor *w = nullptr, *b = nullptr;
    float eps = 1e-5f;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, float eps_ = 1e-5f) {
        eps = eps_;
        w   = ps.add(prefix + "weight", GGML_TYPE_F32, {dim}, InitKind::NORM_SCALE, dim);
        b   = ps.add(prefix + "bias", GGML_TYPE_F32, {dim}, InitKind::BIAS, dim);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ext_layer_norm(g.ctx, x, w, b, eps); }
};

// ResBlock — src/model/common/block.hpp:67-179 (dims == 2)
struct ResBlock {
    int64_t channels = 0, out_channels = 0;
    GroupNorm32 in_norm, out_norm;
    Conv2d in_conv, out_conv, skip;
    Linear emb;
    void init(ParamStore& ps, const std::string& prefix, int64_t ch, int64_t emb_ch, int64_t out_ch) {
        channels     = ch;
        out_channels = out_ch;
        in_norm.init(ps, prefix + "in_layers.0.", ch);
        in_conv.init(ps, prefix + "in_layers.2.", ch, out_ch, 3, 1, 1);
        emb.init(ps, prefix + "emb_layers.1.", emb_ch, out_ch);
        out_norm.init(ps, prefix + "out_layers.0.", out_ch);
        out_conv.init(ps, prefix + "out_layers.3.", out_ch, out_ch, 3, 1, 1);
        if (out_ch != ch) skip.init(ps, prefix + "skip_connection.", ch, out_ch, 1, 1, 0);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* emb_in) const {
        ggml_context* c = g.ctx;
        ggml_tensor* h  = in_norm.forward(g, x);
        h               = ggml_silu_inplace(c, h);
        h               = in_conv.forward(g, h);
        ggml_tensor* e  = ggml_silu(c, emb_in);
        e               = emb.forward(g, e);
        e               = ggml_reshape_4d(c, e, 1, 1, e->ne[0], e->ne[1]);
        h               = ggml_add(c, h, e);
        h               = out_norm.forward(g, h);
        h               = ggml_silu_inplace(c, h);
        h               = out_conv.forward(g, h);
        if (out_channels != channels) x = skip.forward(g, x);
        return ggml_add(c, h, x);
    }
};

// GEGLU + FeedForward — block.hpp:182-304
struct FeedForward {
    Linear proj, out;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, int64_t dim_out, int64_t mult = 4) {
        const int64_t inner = dim * mult;
        proj.init(ps, prefix + "net.0.proj.", dim, inner * 2);
        out.init(ps, prefix + "net.2.", inner, dim_out);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const {
        ggml_context* c   = g.ctx;
        x                 = proj.forward(g, x);
        auto parts        = ext_chunk(c, x, 2, 0, false);
        ggml_tensor* gate = ggml_cont(c, parts[1]);
        gate              = ext_gelu(c, gate, true);
        x                 = ggml_mul(c, parts[0], gate);
        return out.forward(g, x);
    }
};

// CrossAttention — block.hpp:307-393
struct CrossAttention {
    int64_t n_head = 0;
    Linear to_q, to_k, to_v, to_out;
    void init(ParamStore& ps, const std::string& prefix, int64_t query_dim, int64_t context_dim, int64_t heads, int64_t d_head) {
        n_head              = heads;
        const int64_t inner = heads * d_head;
        to_q.init(ps, prefix + "to_q.", query_dim, inner, false);
        to_k.init(ps, prefix + "to_k.", context_dim, inner, false);
        to_v.init(ps, prefix + "to_v.", context_dim, inner, false);
        to_out.init(ps, prefix + "to_out.0.", inner, query_dim);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* context) const {
        ggml_tensor* q = to_q.forward(g, x);
        ggml_tensor* k = to_k.forward(g, context);
        ggml_tensor* v = to_v.forward(g, context);
        x              = ext_attention(g, q, k, v, n_head);
        return to_out.forward(g, x);
    }
};

// BasicTransformerBlock — block.hpp:396-466
struct BasicTransformerBlock {
    CrossAttention attn1, attn2;
    FeedForward ff;
    LayerNorm norm1, norm2, norm3;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, int64_t heads, int64_t d_head, int64_t context_dim) {
        attn1.init(ps, prefix + "attn1.", dim, dim, heads, d_head);
        attn2.init(ps, prefix + "attn2.", dim, context_dim, heads, d_head);
        ff.init(ps, prefix + "ff.", dim, dim);
        norm1.init(ps, prefix + "norm1.", dim);
        norm2.init(ps, prefix + "norm2.", dim);
        norm3.init(ps, prefix + "norm3.", dim);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* context) const {
        ggml_context* c = g.ctx;
        ggml_tensor* r  = x;
        x               = norm1.forward(g, x);
        x               = attn1.forward(g, x, x);
        x               = ggml_add(c, x, r);
        r               = x;
        x               = norm2.forward(g, x);
        x               = attn2.forward(g, x, context);
        x               = ggml_add(c, x, r);
        r               = x;
        x               = norm3.forward(g, x);
        x               = ff.forward(g, x);
        return ggml_add(c, x, r);
    }
};

// SpatialTransformer — block.hpp:469-577
struct SpatialTransformer {
    int64_t in_channels = 0, n_head = 0, d_head = 0;
    bool use_linear = false;
    GroupNorm32 norm;
    Conv2d proj_in_c, proj_out_c;
    Linear proj_in_l, proj_out_l;
    std::vector<BasicTransformerBlock> blocks;
    void init(ParamStore& ps, const std::string& prefix, int64_t ch, int64_t heads, int64_t dh, int64_t depth, int64_t context_dim, bool linear) {
        in_channels         = ch;
        n_head              = heads;
        d_head              = dh;
        use_linear          = linear;
        const int64_t inner = heads * dh;
        norm.init(ps, prefix + "norm.", ch);
        if (linear) {
            proj_in_l.init(ps, prefix + "proj_in.", ch, inner);
        } else {
            proj_in_c.init(ps, prefix + "proj_in.", ch, inner, 1);
        }
        blocks.resize(depth);
        for (int64_t i = 0; i < depth; ++i) blocks[i].init(ps, prefix + "transformer_blocks." + std::to_string(i) + ".", inner, heads, dh, context_dim);
        if (linear) {
            proj_out_l.init(ps, prefix + "proj_out.", inner, ch);
        } else {
            proj_out_c.init(ps, prefix + "proj_out.", inner, ch, 1);
        }
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* context) const {
        ggml_context* c   = g.ctx;
        ggml_tensor* x_in = x;
        const int64_t n = x->ne[3], h = x->ne[1], w = x->ne[0], inner = n_head * d_head;
        x = norm.forward(g, x);
        if (use_linear) {
            x = ggml_cont(c, ggml_permute(c, x, 1, 2, 0, 3));
            x = ggml_reshape_3d(c, x, inner, w * h, n);
            x = proj_in_l.forward(g, x);
        } else {
            x = proj_in_c.forward(g, x);
            x = ggml_cont(c, ggml_permute(c, x, 1, 2, 0, 3));
            x = ggml_reshape_3d(c, x, inner, w * h, n);
        }
        for (auto& b : blocks) x = b.forward(g, x, context);
        if (use_linear) {
            x = proj_out_l.forward(g, x);
            x = ggml_cont(c, ggml_permute(c, x, 1, 0, 2, 3));
            x = ggml_reshape_4d(c, x, w, h, inner, n);
        } else {
            x = ggml_cont(c, ggml_permute(c, x, 1, 0, 2, 3));
            x = ggml_reshape_4d(c, x, w, h, inner, n);
            x = proj_out_c.forward(g, x);
        }
        return ggml_add(c, x, x_in);
    }
};

// ---- DiT building blocks -------------------------------------------------------------------------
// ggml_ext_slice — ggml_extend.hpp:605-638
inline ggml_tensor* ext_slice(ggml_context* c, ggml_tensor* x, int dim, int64_t start, int64_t end, bool cont = true) {
    if (x->ne[dim] == 1) return x;
    while (start < 0) start = x->ne[dim] + start;
    while (end < 0) end = x->ne[dim] + end;
    int64_t ne[4] = {x->ne[0], x->ne[1], x->ne[2], x->ne[3]};
    ne[dim]       = end - start;
    x             = ggml_view_4d(c, x, ne[0], ne[1], ne[2], ne[3], x->nb[1], x->nb[2], x->nb[3], start * x->nb[dim]);
    if (cont) x = ggml_cont(c, x);
    return x;
}

// split_qkv — ggml_extend.hpp:1253-1263: [3C, L, N] -> 3 x [C, L, N] through one permuted copy
inline std::vector<ggml_tensor*> split_qkv(ggml_context* c, ggml_tensor* qkv) {
    qkv = ggml_reshape_4d(c, qkv, qkv->ne[0] / 3, 3, qkv->ne[1], qkv->ne[2]);
    qkv = ggml_cont(c, ggml_permute(c, qkv, 0, 3, 1, 2));
    const size_t off = qkv->nb[2] * qkv->ne[2];
    std::vector<ggml_tensor*> r;
    for (int i = 0; i < 3; ++i) r.push_back(ggml_view_3d(c, qkv, qkv->ne[0], qkv->ne[1], qkv->ne[2], qkv->nb[1], qkv->nb[2], off * i));
    return r;
}

// modulate — src/model/diffusion/mmdit.hpp:368-380: x * (1 + scale) + shift with [C, N] modulation vectors
inline ggml_tensor* modulate(ggml_context* c, ggml_tensor* x, ggml_tensor* shift, ggml_tensor* scale) {
    scale = ggml_reshape_3d(c, scale, scale->ne[0], 1, scale->ne[1]);
    shift = ggml_reshape_3d(c, shift, shift->ne[0], 1, shift->ne[1]);
    x     = ggml_add(c, x, ggml_mul(c, x, scale));
    x     = ggml_add(c, x, shift);
    return x;
}

// RMSNorm — ggml_extend.hpp:3996-4023
struct RMSNorm {
    ggml_tensor* w = nullptr;
    float eps      = 1e-6f;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, float eps_ = 1e-6f) {
        eps = eps_;
        w   = ps.add(prefix + "weight", GGML_TYPE_F32, {dim}, InitKind::NORM_SCALE, dim);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ggml_mul_inplace(g.ctx, ggml_rms_norm(g.ctx, x, eps), w); }
};

// LayerNorm(elementwise_affine = false) — ggml_extend.hpp:3897-3944
struct PlainLayerNorm {
    float eps = 1e-6f;
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ext_layer_norm(g.ctx, x, nullptr, nullptr, eps); }
};

// Rope::apply_rope (interleaved) — src/model/common/rope.hpp:966-1004.  x [d_head, n_head, L, N], pe [2, 2, d_head/2, L] -> [d_head, L, n_head*N]
inline ggml_tensor* apply_rope(ggml_context* c, ggml_tensor* x, ggml_tensor* pe) {
    const int64_t d_head = x->ne[0], n_head = x->ne[1], L = x->ne[2], N = x->ne[3];
    x = ggml_cont(c, ggml_permute(c, x, 0, 2, 1, 3));            // [d_head, L, n_head, N]
    x = ggml_reshape_4d(c, x, 2, d_head / 2, L, n_head * N);
    x = ggml_cont(c, ggml_permute(c, x, 3, 0, 1, 2));            // [d_head/2, L, n_head*N, 2]
    size_t off       = x->nb[2] * x->ne[2];
    ggml_tensor* x_0 = ggml_view_3d(c, x, x->ne[0], x->ne[1], x->ne[2], x->nb[1], x->nb[2], 0);
    ggml_tensor* x_1 = ggml_view_3d(c, x, x->ne[0], x->ne[1], x->ne[2], x->nb[1], x->nb[2], off);
    x_0              = ggml_reshape_4d(c, x_0, 1, x_0->ne[0], x_0->ne[1], x_0->ne[2]);
    x_1              = ggml_reshape_4d(c, x_1, 1, x_1->ne[0], x_1->ne[1], x_1->ne[2]);
    ggml_tensor* tmp = ggml_new_tensor_4d(c, x_0->type, 2, x_0->ne[1], x_0->ne[2], x_0->ne[3]);
    x_0              = ggml_repeat(c, x_0, tmp);
    x_1              = ggml_repeat(c, x_1, tmp);
    pe               = ggml_cont(c, ggml_permute(c, pe, 3, 0, 1, 2));  // [2, d_head/2, L, 2]
    off              = pe->nb[2] * pe->ne[2];
    ggml_tensor* pe_0 = ggml_view_3d(c, pe, pe->ne[0], pe->ne[1], pe->ne[2], pe->nb[1], pe->nb[2], 0);
    ggml_tensor* pe_1 = ggml_view_3d(c, pe, pe->ne[0], pe->ne[1], pe->ne[2], pe->nb[1], pe->nb[2], off);
    ggml_tensor* out  = ggml_add_inplace(c, ggml_mul(c, x_0, pe_0), ggml_mul(c, x_1, pe_1));
    return ggml_reshape_3d(c, out, d_head, L, n_head * N);
}
// Rope::attention — rope.hpp:1006-1025
inline ggml_tensor* rope_attention(GraphCtx& g, ggml_tensor* q, ggml_tensor* k, ggml_tensor* v, ggml_tensor* pe) {
    const int64_t n_head = q->ne[1];
    q = apply_rope(g.ctx, q, pe);
    k = apply_rope(g.ctx, k, pe);
    return ext_attention(g, q, k, v, n_head, true);
}

// Mlp — src/model/common/block.hpp:230-259 (GELU tanh)
struct Mlp {
    Linear fc1, fc2;
    void init(ParamStore& ps, const std::string& prefix, int64_t in, int64_t hidden) {
        fc1.init(ps, prefix + "fc1.", in, hidden);
        fc2.init(ps, prefix + "fc2.", hidden, in);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const {
        x = fc1.forward(g, x);
        x = ext_gelu(g.ctx, x, true);
        return fc2.forward(g, x);
    }
};

}  // namespace sdmi
