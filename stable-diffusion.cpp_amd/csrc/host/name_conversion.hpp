// name_conversion.hpp — checkpoint tensor-name dialects -> the names the graph builders register.
//
// What the reference does (src/name_conversion.cpp:1346-1577 convert_tensor_name and its helpers :35-105 OpenCLIP -> HF CLIP,
// :106-186 cond_stage names, :227-450 diffusers UNet -> original LDM, :906-1000 diffusers VAE -> original): every tensor name read from
// a file is rewritten to ONE canonical dialect — original-LDM names under "model.diffusion_model.", "first_stage_model.",
// "cond_stage_model[.1].transformer.text_model." (UNet families) or "text_encoders.{clip_l,clip_g,t5xxl}.transformer." (DiT families).
// The reference does it with ordered string-replacement tables; here a name is split into path components and the block indices
// are COMPUTED from the architecture (levels, res-blocks per level, which levels carry attention), so the same code serves SD1.x, SDXL and
// the tiny test widths.  Only the dialects of the engine's model families are handled (no LoRA / ControlNet / video models).
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace sdmi {

struct NameDialect {
    bool unet_family = true;  // SD1.x / SDXL (cond_stage_model.* text encoders) vs SD3 / FLUX (text_encoders.*)
    bool flux        = false;
    int unet_levels = 4, unet_res_blocks = 2;
    std::vector<int> unet_attn_levels{0, 1, 2};  // levels (0 = full resolution) whose blocks carry a SpatialTransformer
    int vae_levels = 4;
};

namespace nameconv {

inline bool starts_with(const std::string& s, const std::string& p) { return s.compare(0, p.size(), p) == 0; }

inline std::vector<std::string> split_dots(const std::string& s) {
    std::vector<std::string> parts;
    size_t b = 0;
    while (b <= s.size()) {
        const size_t e = s.find('.', b);
        parts.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return parts;
}
inline std::string join_from(const std::vector<std::string>& p, size_t from) {
    std::string s;
    for (size_t i = from; i < p.size(); ++i) s += (i > from ? "." : "") + p[i];
    return s;
}
inline bool is_index(const std::string& s) { return !s.empty() && s.find_first_not_of("0123456789") == std::string::npos; }
inline bool has_attn(const NameDialect& d, int level) {
    for (int l : d.unet_attn_levels)
        if (l == level) return true;
    return false;
}

// ---- diffusers UNet2DConditionModel -> LDM UNetModel ------------------------------------------------------------------------------
// LDM numbering: input_blocks.0 = conv_in; level i, res-block j -> input_blocks.(i*(R+1)+j+1) with .0 = ResBlock, .1 = transformer; the
// stride-2 conv closing level i -> input_blocks.((i+1)*(R+1)).0.op.  output_blocks.(i*(R+1)+j): .0 ResBlock, .1 transformer (levels
// with attention), the 2x upsample conv of up-block i is the LAST member of output_blocks.(i*(R+1)+R).
inline std::string resnet_member(const std::string& m) {
    if (m == "norm1") return "in_layers.0";
    if (m == "conv1") return "in_layers.2";
    if (m == "norm2") return "out_layers.0";
    if (m == "conv2") return "out_layers.3";
    if (m == "time_emb_proj") return "emb_layers.1";
    if (m == "conv_shortcut") return "skip_connection";
    return m;
}
inline std::string fix_attention_tail(std::string tail) {  // old diffusers wrote to_out without the Sequential index
    for (const char* leaf : {"to_out.weight", "to_out.bias"}) {
        const std::string l = leaf;
        if (tail.size() >= l.size() && tail.compare(tail.size() - l.size(), l.size(), l) == 0) tail.insert(tail.size() - (l.size() - 6), ".0");
    }
    return tail;
}
inline bool is_diffusers_unet(const std::vector<std::string>& p) {
    if (p.empty()) return false;
    const std::string& h = p[0];
    return h == "down_blocks" || h == "up_blocks" || h == "mid_block" || h == "conv_in" || h == "conv_out" || h == "conv_norm_out" || h == "time_embedding" ||
           h == "add_embedding";
}
inline std::string unet_diffusers_to_ldm(const std::string& name, const NameDialect& d) {
    const std::vector<std::string> p = split_dots(name);
    if (!is_diffusers_unet(p)) return name;
    const int R = d.unet_res_blocks;
    const std::string& h = p[0];
    if (h == "conv_in") return "input_blocks.0.0." + join_from(p, 1);
    if (h == "conv_norm_out") return "out.0." + join_from(p, 1);
    if (h == "conv_out") return "out.2." + join_from(p, 1);
    if ((h == "time_embedding" || h == "add_embedding") && p.size() >= 3) {
        const std::string seq = p[1] == "linear_1" ? "0" : (p[1] == "linear_2" ? "2" : p[1]);
        return (h == "time_embedding" ? "time_embed." : "label_emb.0.") + seq + "." + join_from(p, 2);
    }
    if (h == "mid_block" && p.size() >= 4 && is_index(p[2])) {
        const int j = std::stoi(p[2]);
        if (p[1] == "resnets") return "middle_block." + std::to_string(2 * j) + "." + resnet_member(p[3]) + (p.size() > 4 ? "." + join_from(p, 4) : "");
        if (p[1] == "attentions") return "middle_block.1." + fix_attention_tail(join_from(p, 3));
        return name;
    }
    if ((h == "down_blocks" || h == "up_blocks") && p.size() >= 5 && is_index(p[1]) && is_index(p[3])) {
        const int i = std::stoi(p[1]), j = std::stoi(p[3]);
        const bool down = h == "down_blocks";
        // diffusers up_blocks run deepest-first, like LDM output_blocks: up-block i works at level (levels-1-i)
        const int level = down ? i : d.unet_levels - 1 - i;
        const std::string blk = (down ? "input_blocks." : "output_blocks.") + std::to_string(down ? i * (R + 1) + j + 1 : i * (R + 1) + j);
        if (p[2] == "resnets") return blk + ".0." + resnet_member(p[4]) + (p.size() > 5 ? "." + join_from(p, 5) : "");
        if (p[2] == "attentions") return blk + ".1." + fix_attention_tail(join_from(p, 4));
        if (p[2] == "downsamplers" && p[4] == "conv") return "input_blocks." + std::to_string((i + 1) * (R + 1)) + ".0.op." + join_from(p, 5);
        if (p[2] == "upsamplers") return "output_blocks." + std::to_string(i * (R + 1) + R) + "." + (has_attn(d, level) ? "2" : "1") + "." + join_from(p, 4);
    }
    return name;
}

// ---- diffusers AutoencoderKL -> LDM AutoencoderKL ----------------------------------------------------------------------------------
inline std::string vae_attn_member(const std::string& m) {
    if (m == "group_norm") return "norm";
    if (m == "to_q" || m == "query") return "q";
    if (m == "to_k" || m == "key") return "k";
    if (m == "to_v" || m == "value") return "v";
    if (m == "proj_attn") return "proj_out";
    return m;
}
inline std::string vae_diffusers_to_ldm(const std::string& name, const NameDialect& d) {
    std::vector<std::string> p = split_dots(name);
    if (p.size() < 3 || (p[0] != "encoder" && p[0] != "decoder")) return name;
    const std::string side = p[0] + ".";
    auto resnet_tail = [&](size_t from) {
        std::string t = join_from(p, from);
        if (starts_with(t, "conv_shortcut")) t = "nin_shortcut" + t.substr(13);
        return t;
    };
    if (p[1] == "conv_norm_out") return side + "norm_out." + join_from(p, 2);
    // an LDM-layout name whose attention members are spelled the diffusers way (mid.attn_1.to_q / to_k / to_v / to_out.0): the reference rewrites those whenever
    // the result holds "mid.attn_1." (vae_extra_conversion_map, src/name_conversion.cpp:957-996) — found by the table test against its own code (round 5)
    if (p[1] == "mid" && p[2] == "attn_1" && p.size() >= 5) {
        if (p[3] == "to_out" && p.size() >= 6 && p[4] == "0") return side + "mid.attn_1.proj_out." + join_from(p, 5);
        if (p[3] == "to_q" || p[3] == "to_k" || p[3] == "to_v") return side + "mid.attn_1." + vae_attn_member(p[3]) + "." + join_from(p, 4);
    }
    if (p[1] == "mid_block" && p.size() >= 5 && is_index(p[3])) {
        if (p[2] == "resnets") return side + "mid.block_" + std::to_string(std::stoi(p[3]) + 1) + "." + resnet_tail(4);
        if (p[2] == "attentions") {
            // to_out.0.{weight,bias} -> proj_out.{weight,bias}
            if (p[4] == "to_out" && p.size() >= 7 && p[5] == "0") return side + "mid.attn_1.proj_out." + join_from(p, 6);
            return side + "mid.attn_1." + vae_attn_member(p[4]) + (p.size() > 5 ? "." + join_from(p, 5) : "");
        }
    }
    if ((p[1] == "up_blocks" || p[1] == "down_blocks") && p.size() >= 6 && is_index(p[2]) && is_index(p[4])) {
        const int i      = std::stoi(p[2]);
        const bool up    = p[1] == "up_blocks";
        const std::string lvl = side + (up ? "up." + std::to_string(d.vae_levels - 1 - i) : "down." + std::to_string(i)) + ".";
        if (p[3] == "resnets") return lvl + "block." + p[4] + "." + resnet_tail(5);
        if (p[3] == "upsamplers") return lvl + "upsample." + join_from(p, 5);
        if (p[3] == "downsamplers") return lvl + "downsample." + join_from(p, 5);
    }
    return name;
}

// ---- CLIP text tower: OpenCLIP ("model.*") and HF variants -> "transformer.text_model.*" -------------------------------------------
inline std::string clip_to_hf(const std::string& name) {
    std::vector<std::string> p = split_dots(name);
    if (p.size() >= 2 && p[0] == "model") {  // OpenCLIP TextTransformer
        const std::string tm = "transformer.text_model.";
        if (p[1] == "ln_final" && p.size() == 3) return tm + "final_layer_norm." + p[2];
        if (p[1] == "positional_embedding") return tm + "embeddings.position_embedding.weight";
        if (p[1] == "token_embedding" && p.size() == 3) return tm + "embeddings.token_embedding." + p[2];
        if (p[1] == "text_projection") return tm + "text_projection";  // "model.text_projection" and "model.text_projection.weight"
        if (p[1] == "transformer" && p.size() >= 6 && p[2] == "resblocks" && is_index(p[3])) {
            const std::string layer = tm + "encoder.layers." + p[3] + ".";
            const std::string rest  = join_from(p, 4);
            if (rest == "attn.in_proj_weight") return layer + "self_attn.in_proj.weight";
            if (rest == "attn.in_proj_bias") return layer + "self_attn.in_proj.bias";
            if (starts_with(rest, "attn.out_proj.")) return layer + "self_attn.out_proj." + rest.substr(14);
            if (starts_with(rest, "ln_1.")) return layer + "layer_norm1." + rest.substr(5);
            if (starts_with(rest, "ln_2.")) return layer + "layer_norm2." + rest.substr(5);
            if (starts_with(rest, "mlp.c_fc.")) return layer + "mlp.fc1." + rest.substr(9);
            if (starts_with(rest, "mlp.c_proj.")) return layer + "mlp.fc2." + rest.substr(11);
        }
        return name;
    }
    if (name == "transformer.text_projection.weight") return "transformer.text_model.text_projection";
    // HF module trees saved without the wrapper: "text_model.*" -> "transformer.text_model.*"
    if (!p.empty() && p[0] == "text_model") return "transformer." + name;
    if (name == "text_projection.weight") return "transformer.text_model.text_projection";
    return name;
}

// ---- T5: llama.cpp GGUF names ("enc.blk.N.attn_q.weight") -> HF ------------------------------------------------------------------------
inline std::string t5_to_hf(const std::string& name) {
    std::vector<std::string> p = split_dots(name);
    if (!p.empty() && p[0] == "token_embd") return "shared." + join_from(p, 1);
    if (p.size() >= 3 && p[0] == "enc" && p[1] == "output_norm") return "encoder.final_layer_norm." + join_from(p, 2);
    if (p.size() >= 5 && p[0] == "enc" && p[1] == "blk" && is_index(p[2])) {
        static const std::pair<const char*, const char*> leaf[] = {
            {"attn_q", "layer.0.SelfAttention.q"},       {"attn_k", "layer.0.SelfAttention.k"},
            {"attn_v", "layer.0.SelfAttention.v"},       {"attn_o", "layer.0.SelfAttention.o"},
            {"attn_norm", "layer.0.layer_norm"},         {"attn_rel_b", "layer.0.SelfAttention.relative_attention_bias"},
            {"ffn_norm", "layer.1.layer_norm"},          {"ffn_gate", "layer.1.DenseReluDense.wi_0"},
            {"ffn_up", "layer.1.DenseReluDense.wi_1"},   {"ffn_down", "layer.1.DenseReluDense.wo"},
        };
        for (auto& l : leaf)
            if (p[3] == l.first) return "encoder.block." + p[2] + "." + l.second + "." + join_from(p, 4);
    }
    return name;
}

// ---- diffusers SD3Transformer2DModel / FluxTransformer2DModel -> original MMDiT / Flux names ------------------------------------------
// (name_conversion.cpp:452-556 convert_diffusers_dit_to_original_sd3, :558-683 convert_diffusers_dit_to_original_flux).  diffusers keeps
// q / k / v (and FLUX's single-block proj_mlp) as separate Linears; the original models own ONE fused Linear whose output rows are the
// parts in order.  Like the reference, a part other than the first is named "<fused name>.<part index>" ("...attn.qkv.weight.1"): the
// loader (engine.cpp sd_load_weights) stacks "<name>", "<name>.1", "<name>.2", ... along the output dimension.  Names that are already in
// the original dialect pass through unchanged.
inline std::string part_suffix(int part) { return part > 0 ? "." + std::to_string(part) : std::string(); }

inline std::string dit_attn_member(const std::vector<std::string>& p, size_t i, bool flux, bool single, const std::string& attn) {
    // p[i] = to_q | to_k | to_v | add_q_proj | ... | norm_q | norm_k | norm_added_q | norm_added_k | to_out | to_add_out;  attn = "attn" | "attn2"
    const std::string& m = p[i];
    const std::string leaf = join_from(p, i + 1);
    const std::string xs = flux ? (single ? "" : "img_" + attn + ".") : "x_block." + attn + ".";
    const std::string cs = flux ? "txt_" + attn + "." : "context_block." + attn + ".";
    const std::string fused = single ? "linear1." : "qkv.";
    static const char* qkv[3] = {"q", "k", "v"};
    for (int k = 0; k < 3; ++k) {
        if (m == std::string("to_") + qkv[k]) return xs + fused + leaf + part_suffix(k);
        if (m == std::string("add_") + qkv[k] + "_proj") return cs + "qkv." + leaf + part_suffix(k);
    }
    if (m == "norm_q" || m == "norm_k" || m == "norm_added_q" || m == "norm_added_k") {
        const bool added = m.compare(0, 10, "norm_added") == 0, q = m.back() == 'q';
        if (flux) return (added ? cs : xs) + (q ? "norm.query_norm." : "norm.key_norm.") + (leaf == "weight" ? "scale" : leaf);
        return (added ? cs : xs) + (q ? "ln_q." : "ln_k.") + leaf;
    }
    if (m == "to_out" && i + 1 < p.size() && p[i + 1] == "0") return xs + "proj." + join_from(p, i + 2);
    if (m == "to_add_out") return cs + "proj." + leaf;
    return std::string();
}

inline std::string dit_diffusers_to_original(const std::string& name, bool flux) {
    const std::vector<std::string> p = split_dots(name);
    if (p.size() < 2) return name;
    const std::string leaf2 = p.size() >= 2 ? p[p.size() - 1] : std::string();
    if ((p[0] == "time_text_embed" || p[0] == "time_embed") && p.size() == 4) {
        const int which = p[2] == "linear_1" ? 0 : (p[2] == "linear_2" ? 1 : -1);
        if (which < 0) return name;
        if (flux) {
            const std::string tgt = p[1] == "timestep_embedder" ? "time_in." : p[1] == "text_embedder" ? "vector_in." : p[1] == "guidance_embedder" ? "guidance_in." : "";
            return tgt.empty() ? name : tgt + (which ? "out_layer." : "in_layer.") + p[3];
        }
        const std::string tgt = p[1] == "timestep_embedder" ? "t_embedder." : p[1] == "text_embedder" ? "y_embedder." : "";
        return tgt.empty() ? name : tgt + (which ? "mlp.2." : "mlp.0.") + p[3];
    }
    if (!flux && p[0] == "pos_embed" && p.size() >= 2) {
        if (p[1] == "pos_embed") return "pos_embed";
        if (p[1] == "proj") return "x_embedder.proj." + join_from(p, 2);
        return name;
    }
    if (flux && p[0] == "context_embedder") return "txt_in." + join_from(p, 1);
    if (flux && p[0] == "x_embedder") return "img_in." + join_from(p, 1);
    if (p[0] == "proj_out" && p.size() == 2) return "final_layer.linear." + p[1];
    if (p[0] == "norm_out" && p.size() == 3 && p[1] == "linear") return "final_layer.adaLN_modulation.1." + p[2];
    const bool dbl = p[0] == "transformer_blocks", sgl = flux && p[0] == "single_transformer_blocks";
    if ((dbl || sgl) && p.size() >= 4 && is_index(p[1])) {
        const std::string dst = (sgl ? "single_blocks." : flux ? "double_blocks." : "joint_blocks.") + p[1] + ".";
        const std::string& m = p[2];
        std::string r;
        if (sgl) {
            if (m == "norm" && p[3] == "linear") r = "modulation.lin." + join_from(p, 4);
            else if (m == "attn") r = dit_attn_member(p, 3, true, true, "attn");
            else if (m == "proj_mlp") r = "linear1." + join_from(p, 3) + part_suffix(3);
            else if (m == "proj_out") r = "linear2." + join_from(p, 3);
        } else {
            const std::string xb = flux ? "img_" : "x_block.", cb = flux ? "txt_" : "context_block.";
            if ((m == "norm1" || m == "norm1_context") && p[3] == "linear")
                r = (m == "norm1" ? xb : cb) + (flux ? "mod.lin." : "adaLN_modulation.1.") + join_from(p, 4);
            else if (m == "attn" || (m == "attn2" && !flux)) r = dit_attn_member(p, 3, flux, false, m);
            else if ((m == "ff" || m == "ff_context") && p.size() >= 5 && p[3] == "net") {
                const std::string blk = m == "ff" ? xb : cb;
                if (p[4] == "0" && p.size() >= 7 && p[5] == "proj") r = blk + (flux ? "mlp.0." : "mlp.fc1.") + join_from(p, 6);
                else if (p[4] == "2") r = blk + (flux ? "mlp.2." : "mlp.fc2.") + join_from(p, 5);
            }
        }
        return r.empty() ? name : dst + r;
    }
    // original-dialect FLUX files that store the RMSNorm scale as "weight" (name_conversion.cpp:632-636, 668-670)
    if (flux && p.size() >= 3 && p[p.size() - 1] == "weight" && (p[p.size() - 2] == "query_norm" || p[p.size() - 2] == "key_norm") && p[p.size() - 3] == "norm")
        return name.substr(0, name.size() - 6) + "scale";
    return name;
}

}  // namespace nameconv

// file name -> canonical engine name (unchanged when already canonical or not recognised)
inline std::string canonical_tensor_name(const std::string& raw, const NameDialect& d) {
    using namespace nameconv;
    const std::string MDM = "model.diffusion_model.", FSM = "first_stage_model.";
    const std::string L = d.unet_family ? "cond_stage_model." : "text_encoders.clip_l.";
    const std::string G = d.unet_family ? "cond_stage_model.1." : "text_encoders.clip_g.";
    const std::string T = "text_encoders.t5xxl.transformer.";
    // component aliases, longest first where one is a prefix of another (name_conversion.cpp:1454-1488)
    const std::pair<std::string, std::string> alias[] = {
        {MDM, MDM},
        {"diffusion_model.", MDM},
        {"unet.", MDM},
        {FSM, FSM},
        {"vae.", FSM},
        {"conditioner.embedders.0.open_clip.", "cond_stage_model."},
        {"conditioner.embedders.0.", "cond_stage_model."},
        {"conditioner.embedders.1.", "cond_stage_model.1."},
        {"cond_stage_model.1.", "cond_stage_model.1."},
        {"cond_stage_model.", "cond_stage_model."},
        {"text_encoders.clip_l.", "text_encoders.clip_l."},
        {"text_encoders.clip_g.", "text_encoders.clip_g."},
        {T, T},
        {"text_encoders.t5xxl.", "text_encoders.t5xxl."},
        {"text_encoder_2.", G + "transformer."},
        {"text_encoder.2.", G + "transformer."},
        {"text_encoder_3.", T},
        {"text_encoder.", L + "transformer."},
        {"te1.", L + "transformer."},
        {"te2.", G + "transformer."},
        {"te3.", T},
        {"te.", L + "transformer."},
        {"clip_l.", L + "transformer."},
        {"clip_g.", G + "transformer."},
        {"t5xxl.", "text_encoders.t5xxl."},
    };
    std::string prefix, rest = raw;
    for (auto& a : alias)
        if (starts_with(raw, a.first)) {
            prefix = a.second;
            rest   = raw.substr(a.first.size());
            break;
        }
    if (prefix.empty()) {
        // bare component files (diffusers keeps one file per sub-model): recognise the UNet / DiT by its top-level members
        if (starts_with(raw, "transformer.") && !d.unet_family) return MDM + dit_diffusers_to_original(raw.substr(12), d.flux);
        return raw;
    }
    if (prefix == MDM) return prefix + (d.unet_family ? unet_diffusers_to_ldm(rest, d) : dit_diffusers_to_original(rest, d.flux));
    if (prefix == FSM) return prefix + vae_diffusers_to_ldm(rest, d);
    if (prefix == T || prefix == "text_encoders.t5xxl.") {
        if (prefix != T && starts_with(rest, "transformer.")) rest = rest.substr(12);
        return T + t5_to_hf(rest);
    }
    // CLIP towers.  After an alias that already ends in "transformer." the HF tree starts at "text_model."
    if (prefix.size() >= 12 && prefix.compare(prefix.size() - 12, 12, "transformer.") == 0) {
        if (rest == "text_projection.weight") return prefix + "text_model.text_projection";
        return prefix + rest;
    }
    return prefix + clip_to_hf(rest);
}

// "….self_attn.in_proj.{weight,bias}" (OpenCLIP fused qkv) -> the three per-projection names the graph registers; empty if not fused
inline std::vector<std::string> split_in_proj_names(const std::string& name) {
    const std::string key = "self_attn.in_proj.";
    const size_t pos      = name.find(key);
    if (pos == std::string::npos) return {};
    const std::string head = name.substr(0, pos) + "self_attn.", leaf = name.substr(pos + key.size());
    return {head + "q_proj." + leaf, head + "k_proj." + leaf, head + "v_proj." + leaf};
}

}  // namespace sdmi
