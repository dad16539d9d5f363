// text_encoders.hpp — the text-encoder step in front of the denoise path: CLIP text transformers
// (OpenAI ViT-L/14, OpenCLIP ViT-H/14, OpenCLIP ViT-bigG/14) and the T5 encoder stack, as ggml graphs on the same backend.
//
// Node-for-node restatement of src/model/te/clip.hpp:12-123 (CLIPMLP / CLIPLayer / CLIPEncoder), :125-181 (CLIPEmbeddings),
// :231-323 (CLIPTextModel), src/core/ggml_extend.hpp:4025-4096 (MultiheadAttention), :3547-3586 (Embedding) and
// src/model/te/t5.hpp:95-385 (T5LayerNorm … T5), :471-530 (relative position buckets).  Tokenisation is NOT here: the reference's
// vocabularies are stripped from the source drop, so every entry point takes token ids (+ per-token weights).
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "nn.hpp"

namespace sdmi {

// ---- Embedding (ggml_extend.hpp:3547-3586): ids [n_token, N] i32 -> [dim, n_token, N] f32 -------------------------
struct Embedding {
    ggml_tensor* w = nullptr;
    void init(ParamStore& ps, const std::string& prefix, int64_t num, int64_t dim, ggml_type type = GGML_TYPE_F32) {
        w = ps.add(prefix + "weight", type, {dim, num}, InitKind::WEIGHT, dim);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* ids) const {
        const int64_t n = ids->ne[1];
        ids             = ggml_reshape_1d(g.ctx, ids, ids->ne[0] * ids->ne[1]);
        ids             = ggml_reshape_3d(g.ctx, ids, ids->ne[0], 1, ids->ne[1]);
        ggml_tensor* e  = ggml_get_rows(g.ctx, w, ids);
        return ggml_reshape_3d(g.ctx, e, e->ne[0], e->ne[1] / n, n);
    }
};

// the table type rule of CLIPEmbeddings / Embedding (clip.hpp:132-142, ggml_extend.hpp:3539-3556): the file's type when GET_ROWS can
// read it, else f32.  With synthetic weights the "file type" is the context's wtype.
inline ggml_type embedding_table_type(ggml_type wtype, int64_t dim) {
    switch (wtype) {
        case GGML_TYPE_F16:
        case GGML_TYPE_Q8_0:
        case GGML_TYPE_Q4_0: return dim % ggml_blck_size(wtype) == 0 ? wtype : GGML_TYPE_F32;  // whole blocks per row only
        default: return GGML_TYPE_F32;
    }
}

// ---- CLIP ----------------------------------------------------------------------------------------------------------
enum ClipVersion { CLIP_VIT_L_14 = 0, CLIP_VIT_H_14 = 1, CLIP_VIT_BIGG_14 = 2 };

struct ClipTextConfig {  // clip.hpp:245-274
    ClipVersion version       = CLIP_VIT_L_14;
    int64_t vocab_size        = 49408;
    int64_t n_token           = 77;
    int64_t hidden_size       = 768;
    int64_t intermediate_size = 3072;
    int64_t n_head            = 12;
    int64_t n_layer           = 12;
    int64_t projection_dim    = 0;  // > 0: text_projection exists (bigG)
    bool with_final_ln        = true;
    bool use_gelu             = false;  // clip.hpp:21-25: tanh-GELU for the OpenCLIP towers (d_model 1024 / 1280), quick-GELU for ViT-L

    static ClipTextConfig vit_l(bool final_ln) {
        ClipTextConfig c;
        c.with_final_ln = final_ln;
        return c;
    }
    static ClipTextConfig vit_h(bool final_ln) {
        ClipTextConfig c;
        c.version = CLIP_VIT_H_14;
        c.hidden_size = 1024, c.intermediate_size = 4096, c.n_head = 16, c.n_layer = 24;
        c.use_gelu      = true;
        c.with_final_ln = final_ln;
        return c;
    }
    static ClipTextConfig vit_bigg(bool final_ln) {
        ClipTextConfig c;
        c.version = CLIP_VIT_BIGG_14;
        c.hidden_size = 1280, c.intermediate_size = 5120, c.n_head = 20, c.n_layer = 32;
        c.projection_dim = 1280;
        c.use_gelu       = true;
        c.with_final_ln  = final_ln;
        return c;
    }
    // same topology at test width
    static ClipTextConfig tiny(int64_t hidden, int64_t heads, int64_t proj, bool gelu, bool final_ln) {
        ClipTextConfig c;
        c.version = proj > 0 ? CLIP_VIT_BIGG_14 : CLIP_VIT_L_14;
        c.vocab_size = 1000, c.hidden_size = hidden, c.intermediate_size = 2 * hidden, c.n_head = heads, c.n_layer = 3;
        c.projection_dim = proj;
        c.use_gelu       = gelu;
        c.with_final_ln  = final_ln;
        return c;
    }
};

struct MultiheadAttention {  // ggml_extend.hpp:4025-4096 (separate q/k/v projections; the in_proj form is a load-time variant)
    int64_t n_head = 0;
    Linear q_proj, k_proj, v_proj, out_proj;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, int64_t heads) {
        n_head = heads;
        q_proj.init(ps, prefix + "q_proj.", dim, dim);
        k_proj.init(ps, prefix + "k_proj.", dim, dim);
        v_proj.init(ps, prefix + "v_proj.", dim, dim);
        out_proj.init(ps, prefix + "out_proj.", dim, dim);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* mask) const {
        ggml_tensor* q = q_proj.forward(g, x);
        ggml_tensor* k = k_proj.forward(g, x);
        ggml_tensor* v = v_proj.forward(g, x);
        x              = ext_attention(g, q, k, v, n_head, false, mask);
        return out_proj.forward(g, x);
    }
};

struct ClipLayer {  // clip.hpp:44-81
    MultiheadAttention self_attn;
    LayerNorm ln1, ln2;
    Linear fc1, fc2;
    bool use_gelu = false;
    void init(ParamStore& ps, const std::string& prefix, const ClipTextConfig& c) {
        use_gelu = c.use_gelu;
        self_attn.init(ps, prefix + "self_attn.", c.hidden_size, c.n_head);
        ln1.init(ps, prefix + "layer_norm1.", c.hidden_size);
        ln2.init(ps, prefix + "layer_norm2.", c.hidden_size);
        fc1.init(ps, prefix + "mlp.fc1.", c.hidden_size, c.intermediate_size);
        fc2.init(ps, prefix + "mlp.fc2.", c.intermediate_size, c.hidden_size);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* mask) const {
        x              = ggml_add(g.ctx, x, self_attn.forward(g, ln1.forward(g, x), mask));
        ggml_tensor* h = fc1.forward(g, ln2.forward(g, x));
        h              = use_gelu ? ext_gelu(g.ctx, h, true) : ggml_gelu_quick_inplace(g.ctx, h);
        h              = fc2.forward(g, h);
        return ggml_add(g.ctx, x, h);
    }
};

struct ClipTextModel {
    ClipTextConfig cfg;
    ggml_tensor *token_embedding = nullptr, *position_embedding = nullptr, *text_projection = nullptr;
    std::vector<ClipLayer> layers;
    LayerNorm final_ln;

    void init(ParamStore& ps, const std::string& prefix, const ClipTextConfig& c) {
        cfg                = c;
        token_embedding    = ps.add(prefix + "embeddings.token_embedding.weight", embedding_table_type(ps.linear_type, c.hidden_size), {c.hidden_size, c.vocab_size}, InitKind::BIAS, c.hidden_size);
        position_embedding = ps.add(prefix + "embeddings.position_embedding.weight", GGML_TYPE_F32, {c.hidden_size, c.n_token}, InitKind::BIAS, c.hidden_size);
        layers.resize(c.n_layer);
        for (int64_t i = 0; i < c.n_layer; ++i) layers[i].init(ps, prefix + "encoder.layers." + std::to_string(i) + ".", c);
        final_ln.init(ps, prefix + "final_layer_norm.", c.hidden_size);
        // clip.hpp:235-239 declares ne = [projection_dim, hidden_size] (equal for bigG); as a Linear weight it is [in = hidden, out = proj]
        if (c.projection_dim > 0) text_projection = ps.add(prefix + "text_projection", GGML_TYPE_F32, {c.hidden_size, c.projection_dim}, InitKind::WEIGHT, c.hidden_size);
    }

    // ids: [n_token, N] i32; mask: [n_token, n_token] f32 causal.  clip.hpp:286-321
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* ids, ggml_tensor* mask, size_t max_token_idx, bool return_pooled, int clip_skip) const {
        ggml_context* c = g.ctx;
        // CLIPEmbeddings::forward (clip.hpp:160-180)
        ggml_tensor* te;
        if (ids->ne[1] == 1) {
            ggml_tensor* ids3 = ggml_reshape_3d(c, ids, ids->ne[0], 1, ids->ne[1]);
            te                = ggml_get_rows(c, token_embedding, ids3);
            te                = ggml_reshape_3d(c, te, te->ne[0], te->ne[1], te->ne[3]);
        } else {
            // N > 1 (inputs longer than n_token folded into rows, clip.hpp:509-512): GET_ROWS with a batch in ids.ne[2] indexes table
            // planes that a 2-D table does not have; gather over the flattened ids instead, like Embedding::forward (ggml_extend.hpp:3574-3581)
            ggml_tensor* flat = ggml_reshape_3d(c, ggml_reshape_1d(c, ids, ids->ne[0] * ids->ne[1]), ids->ne[0] * ids->ne[1], 1, 1);
            te                = ggml_get_rows(c, token_embedding, flat);
            te                = ggml_reshape_3d(c, te, te->ne[0], ids->ne[0], ids->ne[1]);
        }
        ggml_tensor* x = ggml_add(c, te, position_embedding);
        // CLIPEncoder::forward (clip.hpp:98-122): pooled output always runs every layer
        const int skip = return_pooled ? -1 : clip_skip;
        int layer_idx  = (int)cfg.n_layer - 1;
        if (skip > 0) layer_idx = (int)cfg.n_layer - skip;
        for (int i = 0; i < (int)cfg.n_layer && i <= layer_idx; ++i) x = layers[i].forward(g, x, mask);
        if (return_pooled || cfg.with_final_ln) x = final_ln.forward(g, x);
        if (return_pooled) {
            ggml_tensor* pooled = ggml_view_1d(c, x, cfg.hidden_size, x->nb[1] * max_token_idx);
            if (text_projection) pooled = ext_linear(c, pooled, text_projection, nullptr);
            return pooled;
        }
        return x;
    }

    static std::vector<float> causal_mask(int n) {  // clip.hpp:541-551
        std::vector<float> m((size_t)n * n);
        for (int i0 = 0; i0 < n; ++i0)
            for (int i1 = 0; i1 < n; ++i1) m[(size_t)i1 * n + i0] = i0 > i1 ? -INFINITY : 0.f;
        return m;
    }
};

// ---- T5 encoder ----------------------------------------------------------------------------------------------------
struct T5Config {  // t5.hpp:18-25 (T5-v1.1-XXL encoder)
    int64_t num_layers = 24, model_dim = 4096, inner_dim = 4096, ff_dim = 10240, num_heads = 64, vocab_size = 32128;
    bool relative_attention = true;
    static T5Config xxl() { return T5Config(); }
    static T5Config tiny(int64_t model_dim) {
        T5Config c;
        c.num_layers = 2, c.model_dim = model_dim, c.inner_dim = 64, c.ff_dim = 128, c.num_heads = 4, c.vocab_size = 1000;
        return c;
    }
};

struct T5LayerNorm {  // t5.hpp:95-117: RMS norm, weight multiplied out of place
    ggml_tensor* w = nullptr;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim) { w = ps.add(prefix + "weight", GGML_TYPE_F32, {dim}, InitKind::NORM_SCALE, dim); }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ggml_mul(g.ctx, ggml_rms_norm(g.ctx, x, 1e-6f), w); }
};

struct T5Attention {  // t5.hpp:181-254
    int64_t num_heads = 0, inner_dim = 0;
    Linear q, k, v, o;
    Embedding rel_bias;
    bool has_bias = false;
    void init(ParamStore& ps, const std::string& prefix, const T5Config& c, bool with_bias) {
        num_heads = c.num_heads, inner_dim = c.inner_dim, has_bias = with_bias;
        q.init(ps, prefix + "q.", c.model_dim, c.inner_dim, false);
        k.init(ps, prefix + "k.", c.model_dim, c.inner_dim, false);
        v.init(ps, prefix + "v.", c.model_dim, c.inner_dim, false);
        o.init(ps, prefix + "o.", c.inner_dim, c.model_dim, false);
        if (with_bias) rel_bias.init(ps, prefix + "relative_attention_bias.", 32, c.num_heads, GGML_TYPE_F32);
    }
    // returns {x, past_bias}
    std::pair<ggml_tensor*, ggml_tensor*> forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* past_bias, ggml_tensor* mask, ggml_tensor* buckets) const {
        ggml_context* c = g.ctx;
        ggml_tensor* qq = q.forward(g, x);
        ggml_tensor* kk = k.forward(g, x);
        ggml_tensor* vv = v.forward(g, x);
        if (has_bias && buckets != nullptr) {
            ggml_tensor* values = rel_bias.forward(g, buckets);                       // [heads, L_k, L_q]
            past_bias           = ggml_cont(c, ggml_permute(c, values, 2, 0, 1, 3));  // [L_k, L_q, heads]
        }
        if (past_bias != nullptr) {
            if (mask != nullptr) {
                mask = ggml_repeat(c, mask, past_bias);
                mask = ggml_add(c, mask, past_bias);
            } else {
                mask = past_bias;
            }
        }
        // T5 does not scale the scores: pre-multiply k by sqrt(d_head) to cancel attention's 1/sqrt(d_head) (t5.hpp:247)
        kk = ext_scale(c, kk, sqrtf((float)(inner_dim / num_heads)), true);
        x  = ext_attention(g, qq, kk, vv, num_heads, false, mask);
        return {o.forward(g, x), past_bias};
    }
};

struct T5Block {  // t5.hpp:138-308
    T5Attention attn;
    T5LayerNorm ln0, ln1;
    Linear wi_0, wi_1, wo;
    void init(ParamStore& ps, const std::string& prefix, const T5Config& c, bool with_bias) {
        attn.init(ps, prefix + "layer.0.SelfAttention.", c, with_bias);
        ln0.init(ps, prefix + "layer.0.layer_norm.", c.model_dim);
        wi_0.init(ps, prefix + "layer.1.DenseReluDense.wi_0.", c.model_dim, c.ff_dim, false);
        wi_1.init(ps, prefix + "layer.1.DenseReluDense.wi_1.", c.model_dim, c.ff_dim, false);
        wo.init(ps, prefix + "layer.1.DenseReluDense.wo.", c.ff_dim, c.model_dim, false);
        wo.scale = 1.f / 32.f;  // t5.hpp:143-145: pre-scale against f16 overflow
        ln1.init(ps, prefix + "layer.1.layer_norm.", c.model_dim);
    }
    std::pair<ggml_tensor*, ggml_tensor*> forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* past_bias, ggml_tensor* mask, ggml_tensor* buckets) const {
        ggml_context* c = g.ctx;
        auto r          = attn.forward(g, ln0.forward(g, x), past_bias, mask, buckets);
        x               = ggml_add_inplace(c, r.first, x);
        ggml_tensor* h  = ln1.forward(g, x);
        ggml_tensor* hg = ext_gelu(c, wi_0.forward(g, h), true);
        ggml_tensor* hl = wi_1.forward(g, h);
        h               = wo.forward(g, ggml_mul_inplace(c, hg, hl));
        x               = ggml_add_inplace(c, h, x);
        return {x, r.second};
    }
};

struct T5Model {
    T5Config cfg;
    Embedding shared;
    std::vector<T5Block> blocks;
    T5LayerNorm final_ln;
    void init(ParamStore& ps, const std::string& prefix, const T5Config& c) {
        cfg = c;
        shared.init(ps, prefix + "shared.", c.vocab_size, c.model_dim, embedding_table_type(ps.linear_type, c.model_dim));
        blocks.resize(c.num_layers);
        for (int64_t i = 0; i < c.num_layers; ++i) blocks[i].init(ps, prefix + "encoder.block." + std::to_string(i) + ".", c, !c.relative_attention || i == 0);
        final_ln.init(ps, prefix + "encoder.final_layer_norm.", c.model_dim);
    }
    // ids [n_token, N] i32, buckets [n_token, n_token] i32, mask optional f32 [n_token, 1 | n_token]  (t5.hpp:370-385, 328-351)
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* ids, ggml_tensor* buckets, ggml_tensor* mask) const {
        ggml_tensor* x         = shared.forward(g, ids);
        ggml_tensor* past_bias = nullptr;
        for (auto& b : blocks) {
            auto r    = b.forward(g, x, past_bias, mask, buckets);
            x         = r.first;
            past_bias = r.second;
        }
        return final_ln.forward(g, x);
    }

    // HF T5 `_relative_position_bucket`, bidirectional, 32 buckets, max distance 128 (t5.hpp:471-530): row = query, column = key
    static std::vector<int32_t> relative_position_buckets(int q_len, int k_len) {
        const int nb = 16, max_exact = 8, max_distance = 128;
        std::vector<int32_t> out((size_t)q_len * k_len);
        const float log_base = logf((float)max_distance / max_exact);
        for (int i = 0; i < q_len; ++i)
            for (int j = 0; j < k_len; ++j) {
                const int rel = j - i;
                int b         = rel > 0 ? nb : 0;
                const int a   = rel < 0 ? -rel : rel;
                if (a < max_exact) {
                    b += a;
                } else {
                    const int large = max_exact + (int)((logf((float)a / max_exact) / log_base) * (nb - max_exact));
                    b += std::min(large, nb - 1);
                }
                out[(size_t)i * k_len + j] = b;
            }
        return out;
    }
};

// ---- token weighting (conditioner.hpp:90-125): scale each token row, then restore the chunk's mean -------------------------------
inline void apply_token_weights(std::vector<float>& hidden, int64_t dim, const std::vector<float>& weights) {
    bool all_one = true;
    for (float w : weights) all_one = all_one && w == 1.0f;
    if (all_one || hidden.empty()) return;
    // sd::Tensor::mean accumulates in double over all elements (src/core/tensor.hpp); keep the same order
    auto mean = [&]() {
        double s = 0.0;
        for (float v : hidden) s += v;
        return (float)(s / (double)hidden.size());
    };
    const float m0 = mean();
    for (size_t t = 0; t < weights.size(); ++t)
        for (int64_t d = 0; d < dim; ++d) hidden[t * dim + d] *= weights[t];
    const float m1 = mean();
    if (std::isfinite(m0) && std::isfinite(m1) && m1 != 0.0f) {
        const float r = m0 / m1;
        for (float& v : hidden) v *= r;
    }
}

}  // namespace sdmi
