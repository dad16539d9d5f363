// conditioner.hpp — token ids (+ prompt weights) -> SDCondition {c_crossattn, c_vector}: the host composition around the text-encoder
// graphs.  Restates FrozenCLIPEmbedderWithCustomWords::get_learned_condition_common (src/conditioning/
// conditioner.hpp:414-544: SD1.x / SD2.x / SDXL), SD3CLIPEmbedder (:842-1015) and FluxCLIPEmbedder (:1209-1297).  The encoder
// forwards are injected as callables so this file holds host logic only; engine.cpp binds them to the backend runners.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <string>
#include <vector>

#include "text_encoders.hpp"

namespace sdmi {

struct TokenList {
    std::vector<int32_t> ids;
    std::vector<float> weights;
};

struct Condition {
    std::vector<float> crossattn;  // [n_tokens][ctx_dim] (ggml ne = [ctx_dim, n_tokens])
    int64_t ctx_dim = 0, n_tokens = 0;
    std::vector<float> vec;  // c_vector (y)
};

enum class CondFamily { SD1, SDXL, SD3, FLUX };

struct ConditionerSpec {
    CondFamily family = CondFamily::SD1;
    int64_t l_dim = 768, g_dim = 1280, g_proj = 1280, t5_dim = 4096;
    int32_t eos_id      = 49407;  // CLIPTokenizer EOS (both towers; the bigG tower pads with 0 instead of EOS)
    int64_t adm_dim     = 2816;   // SDXL c_vector size
    int ts_dim          = 256;    // SDXL size/crop embedding width (conditioner.hpp:514)
    size_t clip_chunk   = 77;
    size_t t5_chunk     = 77;  // 77 for SD3 (:857), 256 for FLUX (:1032)
    bool has_l = true, has_g = false, has_t5 = false;
};

struct EncoderFns {
    // which: 0 = clip_l, 1 = clip_g.  out: hidden states [n_tokens][dim] or the pooled vector
    std::function<bool(int which, const std::vector<int32_t>& ids, size_t max_token_idx, bool return_pooled, int clip_skip, std::vector<float>& out)> clip;
    std::function<bool(const std::vector<int32_t>& ids, std::vector<float>& out)> t5;
};

// ggml_extend.hpp:1579-1606 (cos first)
inline void host_timestep_embedding(const std::vector<float>& ts, int dim, std::vector<float>& out, int max_period = 10000) {
    out.assign(ts.size() * dim, 0.f);
    const int half = dim / 2;
    for (size_t i = 0; i < ts.size(); ++i)
        for (int j = 0; j < half; ++j) {
            const float f            = (float)std::exp(-std::log(max_period) * j / half);
            const float arg          = ts[i] * f;
            out[i * dim + j]         = std::cos(arg);
            out[i * dim + j + half]  = std::sin(arg);
        }
}

// rows of a ([n][da]) and b ([n][db]) side by side -> [n][da + db + pad]
inline std::vector<float> concat_features(const std::vector<float>& a, int64_t da, const std::vector<float>& b, int64_t db, int64_t n, int64_t pad_to = 0) {
    const int64_t d = std::max(da + db, pad_to);
    std::vector<float> o((size_t)(n * d), 0.f);
    for (int64_t t = 0; t < n; ++t) {
        std::copy(a.begin() + t * da, a.begin() + (t + 1) * da, o.begin() + t * d);
        std::copy(b.begin() + t * db, b.begin() + (t + 1) * db, o.begin() + t * d + da);
    }
    return o;
}

inline size_t eos_index(const std::vector<int32_t>& chunk, int32_t eos) {
    const size_t d = (size_t)(std::find(chunk.begin(), chunk.end(), eos) - chunk.begin());
    return std::min(d, chunk.size() - 1);
}

inline bool get_learned_condition(const ConditionerSpec& sp, const EncoderFns& fn, const TokenList& clip_l, const TokenList& clip_g, const TokenList& t5,
                                  int clip_skip, int width, int height, bool zero_out_masked, Condition& out, std::string& err) {
    out = Condition();
    auto slice_i = [](const std::vector<int32_t>& v, size_t b, size_t n) { return std::vector<int32_t>(v.begin() + b, v.begin() + b + n); };
    auto slice_f = [](const std::vector<float>& v, size_t b, size_t n) { return std::vector<float>(v.begin() + b, v.begin() + b + n); };
    auto check   = [&](const TokenList& t, size_t chunk, const char* what) {
        if (t.ids.size() != t.weights.size() || t.ids.size() % chunk != 0) {
            err = std::string(what) + ": token count must equal the weight count and be a multiple of the chunk length";
            return false;
        }
        return true;
    };

    if (sp.family == CondFamily::SD1 || sp.family == CondFamily::SDXL) {
        const bool xl = sp.family == CondFamily::SDXL;
        if (!check(clip_l, sp.clip_chunk, "clip tokens") || clip_l.ids.empty()) {
            if (err.empty()) err = "clip tokens: empty";
            return false;
        }
        if (clip_skip <= 0) clip_skip = xl ? 2 : 1;  // :425-427 (SD2.x also uses 2; it is not one of the engine's families)
        const size_t chunks = clip_l.ids.size() / sp.clip_chunk;
        std::vector<float> pooled;
        out.ctx_dim = xl ? sp.l_dim + sp.g_dim : sp.l_dim;
        for (size_t ci = 0; ci < chunks; ++ci) {
            std::vector<int32_t> ids = slice_i(clip_l.ids, ci * sp.clip_chunk, sp.clip_chunk);
            std::vector<float> w     = slice_f(clip_l.weights, ci * sp.clip_chunk, sp.clip_chunk);
            std::vector<int32_t> ids2;
            size_t max_idx = 0;
            if (xl) {  // :440-453: the bigG tower sees zeros after the first EOS
                ids2    = ids;
                auto it = std::find(ids2.begin(), ids2.end(), sp.eos_id);
                if (it != ids2.end()) std::fill(std::next(it), ids2.end(), 0);
                max_idx = eos_index(ids, sp.eos_id);
            }
            std::vector<float> h;
            if (!fn.clip(0, ids, max_idx, false, clip_skip, h)) return false;
            if (xl) {
                std::vector<float> h2;
                if (!fn.clip(1, ids2, max_idx, false, clip_skip, h2)) return false;
                h = concat_features(h, sp.l_dim, h2, sp.g_dim, (int64_t)sp.clip_chunk);
                if (ci == 0 && !fn.clip(1, ids2, max_idx, true, clip_skip, pooled)) return false;
            }
            apply_token_weights(h, out.ctx_dim, w);
            if (zero_out_masked) std::fill(h.begin(), h.end(), 0.f);
            out.crossattn.insert(out.crossattn.end(), h.begin(), h.end());
        }
        out.n_tokens = (int64_t)(chunks * sp.clip_chunk);
        if (xl) {  // :512-534: [pooled | emb(h, w) | emb(0, 0) | emb(h, w)]
            out.vec.assign((size_t)sp.adm_dim, 0.f);
            if ((int64_t)pooled.size() + 6 * sp.ts_dim != sp.adm_dim) {
                err = "SDXL c_vector layout does not add up to adm_in_channels";
                return false;
            }
            size_t off = 0;
            std::copy(pooled.begin(), pooled.end(), out.vec.begin());
            off += pooled.size();
            std::vector<float> e;
            for (const std::vector<float>& ts : {std::vector<float>{(float)height, (float)width}, std::vector<float>{0.f, 0.f}, std::vector<float>{(float)height, (float)width}}) {
                host_timestep_embedding(ts, sp.ts_dim, e);
                std::copy(e.begin(), e.end(), out.vec.begin() + off);
                off += e.size();
            }
        }
        return true;
    }

    if (sp.family == CondFamily::SD3) {
        if (!check(clip_l, sp.clip_chunk, "clip_l tokens") || !check(clip_g, sp.clip_chunk, "clip_g tokens") || !check(t5, sp.clip_chunk, "t5 tokens")) return false;
        if (clip_skip <= 0) clip_skip = 2;
        const size_t L      = sp.clip_chunk;
        const size_t chunks = std::max(std::max(clip_l.ids.size(), clip_g.ids.size()), t5.ids.size()) / L;
        if (chunks == 0) {
            err = "no tokens";
            return false;
        }
        out.ctx_dim = sp.t5_dim;
        for (size_t ci = 0; ci < chunks; ++ci) {
            std::vector<float> hl((size_t)(L * sp.l_dim), 0.f), hg((size_t)(L * sp.g_dim), 0.f), ht((size_t)(L * sp.t5_dim), 0.f), pl, pg;
            if (sp.has_l && clip_l.ids.size() >= (ci + 1) * L) {
                auto ids = slice_i(clip_l.ids, ci * L, L);
                if (!fn.clip(0, ids, 0, false, clip_skip, hl)) return false;
                apply_token_weights(hl, sp.l_dim, slice_f(clip_l.weights, ci * L, L));
                if (ci == 0 && !fn.clip(0, ids, eos_index(ids, sp.eos_id), true, clip_skip, pl)) return false;
            }
            if (sp.has_g && clip_g.ids.size() >= (ci + 1) * L) {
                auto ids = slice_i(clip_g.ids, ci * L, L);
                if (!fn.clip(1, ids, 0, false, clip_skip, hg)) return false;
                apply_token_weights(hg, sp.g_dim, slice_f(clip_g.weights, ci * L, L));
                if (ci == 0 && !fn.clip(1, ids, eos_index(ids, sp.eos_id), true, clip_skip, pg)) return false;
            }
            if (sp.has_t5 && t5.ids.size() >= (ci + 1) * L) {
                if (!fn.t5(slice_i(t5.ids, ci * L, L), ht)) return false;
                apply_token_weights(ht, sp.t5_dim, slice_f(t5.weights, ci * L, L));
            }
            // :983-996: [clip_l | clip_g | zero pad to the T5 width] rows, then the T5 rows
            std::vector<float> lg = concat_features(hl, sp.l_dim, hg, sp.g_dim, (int64_t)L, sp.t5_dim);
            if ((int64_t)(lg.size() / L) != sp.t5_dim) {
                err = "clip_l + clip_g is wider than the T5 stream";
                return false;
            }
            if (zero_out_masked) {
                std::fill(lg.begin(), lg.end(), 0.f);
                std::fill(ht.begin(), ht.end(), 0.f);
            }
            out.crossattn.insert(out.crossattn.end(), lg.begin(), lg.end());
            out.crossattn.insert(out.crossattn.end(), ht.begin(), ht.end());
            if (ci == 0) {
                if (pl.empty()) pl.assign((size_t)sp.l_dim, 0.f);
                if (pg.empty()) pg.assign((size_t)sp.g_proj, 0.f);
                out.vec = pl;
                out.vec.insert(out.vec.end(), pg.begin(), pg.end());
            }
        }
        out.n_tokens = (int64_t)(chunks * 2 * L);
        return true;
    }

    // FLUX (:1209-1297): pooled ViT-L vector from the first 77 tokens, T5 hidden states in chunks of 256
    if (!check(t5, sp.t5_chunk, "t5 tokens")) return false;
    if (clip_skip <= 0) clip_skip = 2;
    const size_t chunks = std::max(clip_l.ids.empty() ? (size_t)0 : sp.t5_chunk, t5.ids.size()) / sp.t5_chunk;
    if (chunks == 0) {
        err = "no tokens";
        return false;
    }
    out.ctx_dim = sp.t5_dim;
    for (size_t ci = 0; ci < chunks; ++ci) {
        if (ci == 0) {
            if (sp.has_l && clip_l.ids.size() >= sp.clip_chunk) {
                auto ids = slice_i(clip_l.ids, 0, sp.clip_chunk);
                if (!fn.clip(0, ids, eos_index(ids, sp.eos_id), true, clip_skip, out.vec)) return false;
            } else {
                out.vec.assign((size_t)sp.l_dim, 0.f);
            }
        }
        std::vector<float> h((size_t)(sp.t5_chunk * sp.t5_dim), 0.f);
        if (sp.has_t5 && t5.ids.size() >= (ci + 1) * sp.t5_chunk) {
            if (!fn.t5(slice_i(t5.ids, ci * sp.t5_chunk, sp.t5_chunk), h)) return false;
            apply_token_weights(h, sp.t5_dim, slice_f(t5.weights, ci * sp.t5_chunk, sp.t5_chunk));
            if (zero_out_masked) std::fill(h.begin(), h.end(), 0.f);
        }
        out.crossattn.insert(out.crossattn.end(), h.begin(), h.end());
    }
    out.n_tokens = (int64_t)(chunks * sp.t5_chunk);
    return true;
}

}  // namespace sdmi
