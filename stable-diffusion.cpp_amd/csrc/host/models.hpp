// models.hpp — UNet (SD1.x / SDXL) and KL-VAE decoder graph builders.
// Topology restated from the reference (cited per struct); emits the reference's node sequences via nn.hpp.
#pragma once
#include <algorithm>

#include "nn.hpp"

namespace sdmi {

// UNetConfig — src/model/diffusion/unet.hpp:16-57
struct UNetConfig {
    bool sdxl                              = false;
    int in_channels                        = 4;
    int out_channels                       = 4;
    int num_res_blocks                     = 2;
    std::vector<int> attention_resolutions = {4, 2, 1};
    std::vector<int> channel_mult          = {1, 2, 4, 4};
    std::vector<int> transformer_depth     = {1, 1, 1, 1};
    int time_embed_dim                     = 1280;
    int num_heads                          = 8;
    int num_head_channels                  = -1;
    int context_dim                        = 768;
    bool use_linear_projection             = false;
    int model_channels                     = 320;
    int adm_in_channels                    = 2816;

    static UNetConfig sd15() { return UNetConfig(); }
    static UNetConfig sdxl_base() {  // unet.hpp:47-57
        UNetConfig c;
        c.sdxl                  = true;
        c.context_dim           = 2048;
        c.attention_resolutions = {4, 2};
        c.channel_mult          = {1, 2, 4};
        c.transformer_depth     = {1, 2, 10};
        c.num_head_channels     = 64;
        c.num_heads             = -1;
        c.use_linear_projection = true;
        return c;
    }
    // reduced-width variant with the SAME topology, for CPU-sized parity tests
    static UNetConfig tiny(bool xl = false) {
        UNetConfig c     = xl ? sdxl_base() : sd15();
        c.model_channels = 32;
        c.time_embed_dim = 128;
        c.context_dim    = 64;
        c.adm_in_channels = 96;
        if (xl) {
            c.num_head_channels = 16;
            c.transformer_depth = {1, 1, 2};
        } else {
            c.num_heads = 2;
        }
        return c;
    }
};

// UnetModelBlock — src/model/diffusion/unet.hpp:298-745
struct UNetModel {
    UNetConfig cfg;
    Linear time_embed_0, time_embed_2, label_emb_0, label_emb_2;
    Conv2d input_conv, out_conv;
    GroupNorm32 out_norm;
    struct Level {
        std::unique_ptr<ResBlock> res;
        std::unique_ptr<SpatialTransformer> attn;
        std::unique_ptr<Conv2d> down;   // DownSampleBlock "op" (block.hpp:8-41)
        std::unique_ptr<Conv2d> up;     // UpSampleBlock "conv" (block.hpp:44-64)
    };
    std::vector<Level> input_blocks;   // index 1.. (0 is input_conv)
    Level mid0, mid1, mid2;
    std::vector<Level> output_blocks;

    bool has_attn(int ds) const { return std::find(cfg.attention_resolutions.begin(), cfg.attention_resolutions.end(), ds) != cfg.attention_resolutions.end(); }
    void heads(int ch, int& n_head, int& d_head) const {
        n_head = cfg.num_heads;
        d_head = ch / std::max(cfg.num_heads, 1);
        if (cfg.num_head_channels != -1) {
            d_head = cfg.num_head_channels;
            n_head = ch / d_head;
        }
    }

    void init(ParamStore& ps, const std::string& prefix, const UNetConfig& c) {
        cfg = c;
        const int mc = cfg.model_channels, ted = cfg.time_embed_dim;
        time_embed_0.init(ps, prefix + "time_embed.0.", mc, ted, true, true);
        time_embed_2.init(ps, prefix + "time_embed.2.", ted, ted, true, true);
        if (cfg.sdxl) {
            label_emb_0.init(ps, prefix + "label_emb.0.0.", cfg.adm_in_channels, ted, true, true);
            label_emb_2.init(ps, prefix + "label_emb.0.2.", ted, ted, true, true);
        }
        input_conv.init(ps, prefix + "input_blocks.0.0.", cfg.in_channels, mc, 3, 1, 1);

        std::vector<int> chans{mc};
        int ch = mc, idx = 0, ds = 1;
        const int L = (int)cfg.channel_mult.size();
        for (int i = 0; i < L; ++i) {
            const int mult = cfg.channel_mult[i];
            for (int j = 0; j < cfg.num_res_blocks; ++j) {
                ++idx;
                Level lv;
                lv.res = std::make_unique<ResBlock>();
                lv.res->init(ps, prefix + "input_blocks." + std::to_string(idx) + ".0.", ch, ted, mult * mc);
                ch = mult * mc;
                if (has_attn(ds)) {
                    int nh, dh;
                    heads(ch, nh, dh);
                    lv.attn = std::make_unique<SpatialTransformer>();
                    lv.attn->init(ps, prefix + "input_blocks." + std::to_string(idx) + ".1.", ch, nh, dh, cfg.transformer_depth[i], cfg.context_dim, cfg.use_linear_projection);
                }
                input_blocks.push_back(std::move(lv));
                chans.push_back(ch);
            }
            if (i != L - 1) {
                ++idx;
                Level lv;
                lv.down = std::make_unique<Conv2d>();
                lv.down->init(ps, prefix + "input_blocks." + std::to_string(idx) + ".0.op.", ch, ch, 3, 2, 1);
                input_blocks.push_back(std::move(lv));
                chans.push_back(ch);
                ds *= 2;
            }
        }
        {
            int nh, dh;
            heads(ch, nh, dh);
            mid0.res = std::make_unique<ResBlock>();
            mid0.res->init(ps, prefix + "middle_block.0.", ch, ted, ch);
            mid1.attn = std::make_unique<SpatialTransformer>();
            mid1.attn->init(ps, prefix + "middle_block.1.", ch, nh, dh, cfg.transformer_depth.back(), cfg.context_dim, cfg.use_linear_projection);
            mid2.res = std::make_unique<ResBlock>();
            mid2.res->init(ps, prefix + "middle_block.2.", ch, ted, ch);
        }
        int oidx = 0;
        for (int i = L - 1; i >= 0; --i) {
            const int mult = cfg.channel_mult[i];
            for (int j = 0; j < cfg.num_res_blocks + 1; ++j) {
                const int ich = chans.back();
                chans.pop_back();
                Level lv;
                lv.res = std::make_unique<ResBlock>();
                lv.res->init(ps, prefix + "output_blocks." + std::to_string(oidx) + ".0.", ch + ich, ted, mult * mc);
                ch         = mult * mc;
                int up_idx = 1;
                if (has_attn(ds)) {
                    int nh, dh;
                    heads(ch, nh, dh);
                    lv.attn = std::make_unique<SpatialTransformer>();
                    lv.attn->init(ps, prefix + "output_blocks." + std::to_string(oidx) + ".1.", ch, nh, dh, cfg.transformer_depth[i], cfg.context_dim, cfg.use_linear_projection);
                    ++up_idx;
                }
                if (i > 0 && j == cfg.num_res_blocks) {
                    lv.up = std::make_unique<Conv2d>();
                    lv.up->init(ps, prefix + "output_blocks." + std::to_string(oidx) + "." + std::to_string(up_idx) + ".conv.", ch, ch, 3, 1, 1);
                    ds /= 2;
                }
                output_blocks.push_back(std::move(lv));
                ++oidx;
            }
        }
        out_norm.init(ps, prefix + "out.0.", ch);
        out_conv.init(ps, prefix + "out.2.", mc, cfg.out_channels, 3, 1, 1);
    }

    // forward — unet.hpp:526-745.  x [W,H,C,N]; timesteps [N]; context [ctx_dim,77,N|1]; y [adm,N|1]
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* timesteps, ggml_tensor* context, ggml_tensor* y) const {
        ggml_context* c = g.ctx;
        if (context != nullptr && context->ne[2] != x->ne[3]) {
            context = ggml_repeat(c, context, ggml_new_tensor_3d(c, GGML_TYPE_F32, context->ne[0], context->ne[1], x->ne[3]));
        }
        if (y != nullptr && y->ne[1] != x->ne[3]) {
            y = ggml_repeat(c, y, ggml_new_tensor_2d(c, GGML_TYPE_F32, y->ne[0], x->ne[3]));
        }
        // ggml_ext_timestep_embedding (ggml_extend.hpp:1644-1652): scale by time_factor 1.0, then embed
        ggml_tensor* t_emb = ggml_timestep_embedding(c, ext_scale(c, timesteps, 1.0f), cfg.model_channels, 10000);
        ggml_tensor* emb   = time_embed_0.forward(g, t_emb);
        emb                = ggml_silu_inplace(c, emb);
        emb                = time_embed_2.forward(g, emb);
        if (y != nullptr) {
            ggml_tensor* le = label_emb_0.forward(g, y);
            le              = ggml_silu_inplace(c, le);
            le              = label_emb_2.forward(g, le);
            emb             = ggml_add(c, emb, le);
        }
        std::vector<ggml_tensor*> hs;
        ggml_tensor* h = input_conv.forward(g, x);
        ggml_set_name(h, "bench-start");
        hs.push_back(h);
        for (auto& lv : input_blocks) {
            if (lv.down) {
                h = lv.down->forward(g, h);
            } else {
                h = lv.res->forward(g, h, emb);
                if (lv.attn) h = lv.attn->forward(g, h, context);
            }
            hs.push_back(h);
        }
        h = mid0.res->forward(g, h, emb);
        h = mid1.attn->forward(g, h, context);
        h = mid2.res->forward(g, h, emb);
        for (auto& lv : output_blocks) {
            ggml_tensor* skip = hs.back();
            hs.pop_back();
            h = ggml_concat(c, h, skip, 2);
            h = lv.res->forward(g, h, emb);
            if (lv.attn) h = lv.attn->forward(g, h, context);
            if (lv.up) {
                h = ggml_upscale(c, h, 2, GGML_SCALE_MODE_NEAREST);
                h = lv.up->forward(g, h);
            }
        }
        h = out_norm.forward(g, h);
        h = ggml_silu_inplace(c, h);
        h = out_conv.forward(g, h);
        ggml_set_name(h, "bench-end");
        return h;
    }
};

// ---- KL-VAE decoder — src/model/vae/auto_encoder_kl.hpp:10-160 (ResnetBlock, AttnBlock), :360-492 (Decoder), :589-620
struct VaeResnetBlock {
    int64_t in_ch = 0, out_ch = 0;
    GroupNorm32 norm1, norm2;
    Conv2d conv1, conv2, nin;
    void init(ParamStore& ps, const std::string& prefix, int64_t ic, int64_t oc) {
        in_ch  = ic;
        out_ch = oc;
        norm1.init(ps, prefix + "norm1.", ic);
        conv1.init(ps, prefix + "conv1.", ic, oc, 3, 1, 1);
        norm2.init(ps, prefix + "norm2.", oc);
        conv2.init(ps, prefix + "conv2.", oc, oc, 3, 1, 1);
        if (ic != oc) nin.init(ps, prefix + "nin_shortcut.", ic, oc, 1, 1, 0);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const {
        ggml_context* c = g.ctx;
        ggml_tensor* h  = norm1.forward(g, x);
        h               = ggml_silu_inplace(c, h);
        h               = conv1.forward(g, h);
        h               = norm2.forward(g, h);
        h               = ggml_silu_inplace(c, h);
        h               = conv2.forward(g, h);
        if (in_ch != out_ch) x = nin.forward(g, x);
        return ggml_add(c, h, x);
    }
};

struct VaeAttnBlock {  // conv (1x1) projections, as SD1.x/SDXL checkpoints store them (auto_encoder_kl.hpp:62-159)
    GroupNorm32 norm;
    Conv2d q, k, v, proj_out;
    void init(ParamStore& ps, const std::string& prefix, int64_t ch) {
        norm.init(ps, prefix + "norm.", ch);
        q.init(ps, prefix + "q.", ch, ch, 1);
        k.init(ps, prefix + "k.", ch, ch, 1);
        v.init(ps, prefix + "v.", ch, ch, 1);
        proj_out.init(ps, prefix + "proj_out.", ch, ch, 1);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const {
        ggml_context* cx = g.ctx;
        ggml_tensor* h_  = norm.forward(g, x);
        const int64_t n = h_->ne[3], c = h_->ne[2], h = h_->ne[1], w = h_->ne[0];
        auto tok = [&](const Conv2d& proj) {
            ggml_tensor* t = proj.forward(g, h_);
            t              = ggml_cont(cx, ggml_permute(cx, t, 1, 2, 0, 3));
            return ggml_reshape_3d(cx, t, c, h * w, n);
        };
        ggml_tensor* qq = tok(q);
        ggml_tensor* kk = tok(k);
        ggml_tensor* vv = tok(v);
        h_              = ext_attention(g, qq, kk, vv, 1);
        h_              = ggml_cont(cx, ggml_permute(cx, h_, 1, 0, 2, 3));
        h_              = ggml_reshape_4d(cx, h_, w, h, c, n);
        h_              = proj_out.forward(g, h_);
        return ggml_add(cx, h_, x);
    }
};

struct VaeConfig {
    int ch                   = 128;
    int out_ch               = 3;
    std::vector<int> ch_mult = {1, 2, 4, 4};
    int num_res_blocks       = 2;
    int z_channels           = 4;
    bool use_quant           = true;   // post_quant_conv present (SD1/SDXL; auto_encoder_kl.hpp:609-612)
    float scale_factor       = 0.18215f;  // auto_encoder_kl.hpp:676-687
    float shift_factor       = 0.f;
    static VaeConfig sd15() { return VaeConfig(); }
    static VaeConfig sdxl() {
        VaeConfig c;
        c.scale_factor = 0.13025f;
        return c;
    }
    static VaeConfig tiny() {
        VaeConfig c;
        c.ch = 32;
        return c;
    }
};

struct VaeDecoder {
    VaeConfig cfg;
    Conv2d post_quant, conv_in, conv_out;
    VaeResnetBlock mid1, mid2;
    VaeAttnBlock mid_attn;
    struct Up {
        std::vector<VaeResnetBlock> blocks;
        std::unique_ptr<Conv2d> upsample;
    };
    std::vector<Up> ups;  // indexed by resolution i (processed from last to first)
    GroupNorm32 norm_out;

    void init(ParamStore& ps, const std::string& prefix, const VaeConfig& c) {
        cfg           = c;
        const int nr  = (int)cfg.ch_mult.size();
        int block_in  = cfg.ch * cfg.ch_mult[nr - 1];
        if (cfg.use_quant) post_quant.init(ps, prefix + "post_quant_conv.", cfg.z_channels, cfg.z_channels, 1);
        const std::string d = prefix + "decoder.";
        conv_in.init(ps, d + "conv_in.", cfg.z_channels, block_in, 3, 1, 1);
        mid1.init(ps, d + "mid.block_1.", block_in, block_in);
        mid_attn.init(ps, d + "mid.attn_1.", block_in);
        mid2.init(ps, d + "mid.block_2.", block_in, block_in);
        ups.resize(nr);
        for (int i = nr - 1; i >= 0; --i) {
            const int block_out = cfg.ch * cfg.ch_mult[i];
            ups[i].blocks.resize(cfg.num_res_blocks + 1);
            for (int j = 0; j < cfg.num_res_blocks + 1; ++j) {
                ups[i].blocks[j].init(ps, d + "up." + std::to_string(i) + ".block." + std::to_string(j) + ".", block_in, block_out);
                block_in = block_out;
            }
            if (i != 0) {
                ups[i].upsample = std::make_unique<Conv2d>();
                ups[i].upsample->init(ps, d + "up." + std::to_string(i) + ".upsample.conv.", block_in, block_in, 3, 1, 1);
            }
        }
        norm_out.init(ps, d + "norm_out.", block_in);
        conv_out.init(ps, d + "conv_out.", block_in, cfg.out_ch, 3, 1, 1);
    }

    // AutoEncoderKL::set_conv2d_scale (auto_encoder_kl.hpp:708-717): every Conv2d block of the autoencoder gets the factor (the reference sets 1/32 for SDXL
    // when no external VAE is given, src/stable-diffusion.cpp:1477-1485: the f16 im2col of ggml-cpu overflows on the SDXL VAE's activations otherwise)
    void set_conv2d_scale(float s) {
        std::vector<Conv2d*> all{&post_quant, &conv_in, &conv_out, &mid1.conv1, &mid1.conv2, &mid1.nin, &mid2.conv1, &mid2.conv2, &mid2.nin,
                                 &mid_attn.q, &mid_attn.k, &mid_attn.v, &mid_attn.proj_out};
        for (auto& u : ups) {
            for (auto& b : u.blocks) {
                all.push_back(&b.conv1);
                all.push_back(&b.conv2);
                all.push_back(&b.nin);
            }
            if (u.upsample) all.push_back(u.upsample.get());
        }
        for (Conv2d* cv : all) cv->scale = s;
    }

    // AutoEncoderKLModel::decode + Decoder::forward
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* z) const {
        ggml_context* c = g.ctx;
        if (cfg.use_quant) z = post_quant.forward(g, z);
        ggml_set_name(z, "bench-start");
        ggml_tensor* h = conv_in.forward(g, z);
        h              = mid1.forward(g, h);
        h              = mid_attn.forward(g, h);
        h              = mid2.forward(g, h);
        for (int i = (int)ups.size() - 1; i >= 0; --i) {
            for (auto& b : ups[i].blocks) h = b.forward(g, h);
            if (ups[i].upsample) {
                h = ggml_upscale(c, h, 2, GGML_SCALE_MODE_NEAREST);
                h = ups[i].upsample->forward(g, h);
            }
        }
        h = norm_out.forward(g, h);
        h = ggml_silu_inplace(c, h);
        h = conv_out.forward(g, h);
        ggml_set_name(h, "bench-end");
        return h;
    }
};

// ---- KL-VAE encoder — src/model/vae/auto_encoder_kl.hpp:276-366 (Encoder), :637-664 (AutoEncoderKLModel::encode: + quant_conv), DownSampleBlock with vae_downsample
// (src/model/common/block.hpp:10-41: pad right / bottom by one, 3x3 stride-2 conv without padding).  Output: the moments [w/8, h/8, 2 * z_channels, N] (mean | log-variance).
struct VaeEncoder {
    VaeConfig cfg;
    Conv2d conv_in, conv_out, quant;
    VaeResnetBlock mid1, mid2;
    VaeAttnBlock mid_attn;
    struct Down {
        std::vector<VaeResnetBlock> blocks;
        std::unique_ptr<Conv2d> downsample;
    };
    std::vector<Down> downs;
    GroupNorm32 norm_out;

    void init(ParamStore& ps, const std::string& prefix, const VaeConfig& c) {
        cfg                 = c;
        const int nr        = (int)cfg.ch_mult.size();
        const std::string e = prefix + "encoder.";
        conv_in.init(ps, e + "conv_in.", 3, cfg.ch, 3, 1, 1);
        int block_in = cfg.ch;
        downs.resize(nr);
        for (int i = 0; i < nr; ++i) {
            block_in            = i == 0 ? cfg.ch : cfg.ch * cfg.ch_mult[i - 1];
            const int block_out = cfg.ch * cfg.ch_mult[i];
            downs[i].blocks.resize(cfg.num_res_blocks);
            for (int j = 0; j < cfg.num_res_blocks; ++j) {
                downs[i].blocks[j].init(ps, e + "down." + std::to_string(i) + ".block." + std::to_string(j) + ".", block_in, block_out);
                block_in = block_out;
            }
            if (i != nr - 1) {
                downs[i].downsample = std::make_unique<Conv2d>();
                downs[i].downsample->init(ps, e + "down." + std::to_string(i) + ".downsample.conv.", block_in, block_in, 3, 2, 0);
            }
        }
        mid1.init(ps, e + "mid.block_1.", block_in, block_in);
        mid_attn.init(ps, e + "mid.attn_1.", block_in);
        mid2.init(ps, e + "mid.block_2.", block_in, block_in);
        norm_out.init(ps, e + "norm_out.", block_in);
        conv_out.init(ps, e + "conv_out.", block_in, 2 * cfg.z_channels, 3, 1, 1);
        if (cfg.use_quant) quant.init(ps, prefix + "quant_conv.", 2 * cfg.z_channels, 2 * cfg.z_channels, 1);
    }
    void set_conv2d_scale(float s) {
        std::vector<Conv2d*> all{&conv_in, &conv_out, &quant, &mid1.conv1, &mid1.conv2, &mid1.nin, &mid2.conv1, &mid2.conv2, &mid2.nin,
                                 &mid_attn.q, &mid_attn.k, &mid_attn.v, &mid_attn.proj_out};
        for (auto& d : downs) {
            for (auto& b : d.blocks) {
                all.push_back(&b.conv1);
                all.push_back(&b.conv2);
                all.push_back(&b.nin);
            }
            if (d.downsample) all.push_back(d.downsample.get());
        }
        for (Conv2d* cv : all) cv->scale = s;
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const {
        ggml_context* c = g.ctx;
        ggml_tensor* h  = conv_in.forward(g, x);
        for (auto& d : downs) {
            for (auto& b : d.blocks) h = b.forward(g, h);
            if (d.downsample) {
                h = ggml_pad_ext(c, h, 0, 1, 0, 1, 0, 0, 0, 0);  // ggml_ext_pad(x, 1, 1): right / bottom
                h = d.downsample->forward(g, h);
            }
        }
        h = mid1.forward(g, h);
        h = mid_attn.forward(g, h);
        h = mid2.forward(g, h);
        h = norm_out.forward(g, h);
        h = ggml_silu_inplace(c, h);
        h = conv_out.forward(g, h);
        if (cfg.use_quant) h = quant.forward(g, h);
        return h;
    }
};

// ---- TAESD: the tiny autoencoder's decoder (SURVEY.md section 8 row f4 "TAESD ... adjacent graphs") — src/model/vae/tae.hpp:15-76 (TAEBlock), :123-183 (TinyDecoder),
// :686-730 (TAESD), :732-792 (TinyImageAutoEncoder: latents enter unscaled, the output is the image in [0, 1] as it leaves the graph).  64 channels throughout:
// tanh(z / 3) * 3 -> conv -> ReLU -> 3 x [3 blocks, nearest x2, bias-free conv] -> block -> conv to RGB.  Sequential indices as the checkpoint keys have them.
struct TaeBlock {
    Conv2d c0, c2, c4;
    void init(ParamStore& ps, const std::string& prefix, int64_t ch) {
        c0.init(ps, prefix + "conv.0.", ch, ch, 3, 1, 1);
        c2.init(ps, prefix + "conv.2.", ch, ch, 3, 1, 1);
        c4.init(ps, prefix + "conv.4.", ch, ch, 3, 1, 1);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const {
        ggml_context* c = g.ctx;
        ggml_tensor* h  = c0.forward(g, x);
        h               = ggml_relu_inplace(c, h);
        h               = c2.forward(g, h);
        h               = ggml_relu_inplace(c, h);
        h               = c4.forward(g, h);
        h               = ggml_add(c, h, x);
        return ggml_relu_inplace(c, h);
    }
};
struct TaeDecoder {
    static constexpr int kChannels = 64, kBlocks = 3;
    int64_t z_channels = 4;
    // layer index -> what sits there (tae.hpp:130-158): 0 conv, 1 ReLU, 2-4 blocks, 5 upsample, 6 conv, 7-9 blocks, 10 upsample, 11 conv, 12-14 blocks, 15 upsample, 16 conv,
    // 17 block, 18 conv
    Conv2d conv_in, conv_out, up_conv[3];
    TaeBlock blocks[3 * kBlocks + 1];
    void init(ParamStore& ps, const std::string& prefix, int64_t zc) {
        z_channels = zc;
        int index  = 0, nb = 0;
        conv_in.init(ps, prefix + std::to_string(index++) + ".", zc, kChannels, 3, 1, 1);
        index++;  // ReLU
        for (int stage = 0; stage < 3; ++stage) {
            for (int i = 0; i < kBlocks; ++i) blocks[nb++].init(ps, prefix + std::to_string(index++) + ".", kChannels);
            index++;  // Upsample
            up_conv[stage].init(ps, prefix + std::to_string(index++) + ".", kChannels, kChannels, 3, 1, 1, /*bias*/ false);
        }
        blocks[nb++].init(ps, prefix + std::to_string(index++) + ".", kChannels);
        conv_out.init(ps, prefix + std::to_string(index++) + ".", kChannels, 3, 3, 1, 1);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* z) const {
        ggml_context* c = g.ctx;
        ggml_tensor* h  = ext_scale(c, z, 1.0f / 3.0f);
        h               = ggml_tanh_inplace(c, h);
        h               = ext_scale(c, h, 3.0f);
        h               = conv_in.forward(g, h);
        h               = ggml_relu_inplace(c, h);
        int nb          = 0;
        for (int stage = 0; stage < 3; ++stage) {
            for (int i = 0; i < kBlocks; ++i) h = blocks[nb++].forward(g, h);
            h = ggml_upscale(c, h, 2, GGML_SCALE_MODE_NEAREST);
            h = up_conv[stage].forward(g, h);
        }
        h = blocks[nb++].forward(g, h);
        return conv_out.forward(g, h);
    }
};

// =====================================================================================================
// MMDiT (SD3 / SD3.5) — src/model/diffusion/mmdit.hpp
// =====================================================================================================
// MMDiTConfig — mmdit.hpp:16-136 (what detect_from_weights derives for SD3.5-large: 38 joint blocks, hidden 64*depth = 2432,
// heads = depth, rms qk-norm, no MMDiT-X self-attention blocks)
struct MMDiTConfig {
    int patch_size             = 2;
    int64_t in_channels        = 16;
    int64_t d_self             = -1;  // >= 0 for MMDiT-X (SD3.5-medium)
    int64_t depth              = 24;
    float mlp_ratio            = 4.0f;
    int64_t adm_in_channels    = 2048;
    int64_t out_channels       = 16;
    int64_t pos_embed_max_size = 192;
    int64_t context_size       = 4096;
    int64_t hidden_size        = 1536;
    bool qk_rms                = false;
    int64_t num_heads          = 0;  // 0: heads = depth, as the reference wires it (mmdit.hpp:798-799)

    static MMDiTConfig sd35_large() {
        MMDiTConfig c;
        c.depth       = 38;
        c.hidden_size = 64 * 38;
        c.qk_rms      = true;
        return c;
    }
    // SD3.5-large's real width (hidden 2432, 38 heads x 64) with TWO joint blocks (one full, one whose context stream is pre_only):
    // full-width block parity against the CPU oracle in seconds (tests/test_zz_gpu_fullsize.py)
    static MMDiTConfig sd35_wide2() {
        MMDiTConfig c = sd35_large();
        c.depth       = 2;
        c.num_heads   = 38;
        return c;
    }
    static MMDiTConfig sd35_wide8() {  // 8 joint blocks at the real width (depth sweep of the full-depth parity test)
        MMDiTConfig c = sd35_large();
        c.depth       = 8;
        c.num_heads   = 38;
        return c;
    }
    // same topology at CPU-test size (last block pre_only, rms qk-norm, one MMDiT-X block to cover that path)
    static MMDiTConfig tiny() {
        MMDiTConfig c;
        c.depth              = 3;
        c.hidden_size        = 64 * 3;
        c.qk_rms             = true;
        c.adm_in_channels    = 64;
        c.context_size       = 96;
        c.pos_embed_max_size = 24;
        c.d_self             = 0;
        return c;
    }
    // SD3-medium's variant at test size: no qk-norm (the q / k parts of the fused projection go to the attention as they are), no MMDiT-X block
    static MMDiTConfig tiny_medium() {
        MMDiTConfig c = tiny();
        c.qk_rms      = false;
        c.d_self      = -1;
        return c;
    }
};

// mmdit.hpp:299-366
struct DitSelfAttention {
    int64_t num_heads = 0;
    bool pre_only     = false, qk_rms = false;
    Linear qkv, proj;
    RMSNorm ln_q, ln_k;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, int64_t heads, bool rms, bool pre_only_) {
        num_heads = heads;
        pre_only  = pre_only_;
        qk_rms    = rms;
        qkv.init(ps, prefix + "qkv.", dim, dim * 3, true);
        if (!pre_only) proj.init(ps, prefix + "proj.", dim, dim);
        if (rms) {
            ln_q.init(ps, prefix + "ln_q.", dim / heads);
            ln_k.init(ps, prefix + "ln_k.", dim / heads);
        }
    }
    std::vector<ggml_tensor*> pre_attention(GraphCtx& g, ggml_tensor* x) const {
        ggml_context* c = g.ctx;
        auto v3         = split_qkv(c, qkv.forward(g, x));
        const int64_t hd = v3[0]->ne[0] / num_heads;
        ggml_tensor* q  = ggml_reshape_4d(c, v3[0], hd, num_heads, v3[0]->ne[1], v3[0]->ne[2]);
        ggml_tensor* k  = ggml_reshape_4d(c, v3[1], hd, num_heads, v3[1]->ne[1], v3[1]->ne[2]);
        if (qk_rms) {
            q = ln_q.forward(g, q);
            k = ln_k.forward(g, k);
        }
        q = ggml_reshape_3d(c, q, q->ne[0] * q->ne[1], q->ne[2], q->ne[3]);
        k = ggml_reshape_3d(c, k, k->ne[0] * k->ne[1], k->ne[2], k->ne[3]);
        return {q, k, v3[2]};
    }
};

// DismantledBlock — mmdit.hpp:382-612
struct DismantledBlock {
    int64_t num_heads = 0;
    bool pre_only = false, self_attn = false;
    PlainLayerNorm norm1, norm2;
    DitSelfAttention attn, attn2;
    Mlp mlp;
    Linear adaLN;
    void init(ParamStore& ps, const std::string& prefix, int64_t hidden, int64_t heads, float mlp_ratio, bool rms, bool pre_only_, bool self_attn_) {
        num_heads = heads;
        pre_only  = pre_only_;
        self_attn = self_attn_;
        attn.init(ps, prefix + "attn.", hidden, heads, rms, pre_only);
        if (self_attn) attn2.init(ps, prefix + "attn2.", hidden, heads, rms, false);
        if (!pre_only) mlp.init(ps, prefix + "mlp.", hidden, (int64_t)(hidden * mlp_ratio));
        const int64_t n_mods = self_attn ? 9 : (pre_only ? 2 : 6);
        adaLN.init(ps, prefix + "adaLN_modulation.1.", hidden, n_mods * hidden);
    }
    struct Pre {
        std::vector<ggml_tensor*> qkv, qkv2, inter;  // inter: x, gate_msa, shift_mlp, scale_mlp, gate_mlp[, gate_msa2]
    };
    Pre pre_attention(GraphCtx& g, ggml_tensor* x, ggml_tensor* cvec) const {
        ggml_context* c = g.ctx;
        const int n_mods = self_attn ? 9 : (pre_only ? 2 : 6);
        ggml_tensor* m   = adaLN.forward(g, ggml_silu(c, cvec));
        auto mv          = ext_chunk(c, m, n_mods, 0, true);
        Pre r;
        if (self_attn) {  // pre_attention_x, mmdit.hpp:424-457
            ggml_tensor* xn = norm1.forward(g, x);
            r.qkv           = attn.pre_attention(g, modulate(c, xn, mv[0], mv[1]));
            r.qkv2          = attn2.pre_attention(g, modulate(c, xn, mv[6], mv[7]));
            r.inter         = {x, mv[2], mv[3], mv[4], mv[5], mv[8]};
        } else {
            r.qkv = attn.pre_attention(g, modulate(c, norm1.forward(g, x), mv[0], mv[1]));
            if (!pre_only) r.inter = {x, mv[2], mv[3], mv[4], mv[5]};
        }
        return r;
    }
    ggml_tensor* post_attention(GraphCtx& g, ggml_tensor* attn_out, ggml_tensor* attn2_out, const std::vector<ggml_tensor*>& it) const {
        ggml_context* c = g.ctx;
        ggml_tensor* x  = it[0];
        auto gate3      = [&](ggml_tensor* t) { return ggml_reshape_3d(c, t, t->ne[0], 1, t->ne[1]); };
        ggml_tensor *gate_msa = gate3(it[1]), *gate_mlp = gate3(it[4]);
        attn_out = attn.proj.forward(g, attn_out);
        if (self_attn) {
            ggml_tensor* gate_msa2 = gate3(it[5]);
            attn2_out              = attn2.proj.forward(g, attn2_out);
            x                      = ggml_add(c, x, ggml_mul(c, attn_out, gate_msa));
            x                      = ggml_add(c, x, ggml_mul(c, attn2_out, gate_msa2));
        } else {
            x = ggml_add(c, x, ggml_mul(c, attn_out, gate_msa));
        }
        ggml_tensor* mlp_out = mlp.forward(g, modulate(c, norm2.forward(g, x), it[2], it[3]));
        return ggml_add(c, x, ggml_mul(c, mlp_out, gate_mlp));
    }
};

struct MMDiTModel {
    MMDiTConfig cfg;
    Conv2d x_embedder;
    Linear t_mlp0, t_mlp2, y_mlp0, y_mlp2, context_embedder, final_linear, final_adaLN;
    PlainLayerNorm norm_final;
    ggml_tensor* pos_embed = nullptr;
    struct Joint {
        DismantledBlock context_block, x_block;
    };
    std::vector<Joint> blocks;

    void init(ParamStore& ps, const std::string& prefix, const MMDiTConfig& c) {
        cfg = c;
        x_embedder.init(ps, prefix + "x_embedder.proj.", cfg.in_channels, cfg.hidden_size, cfg.patch_size, cfg.patch_size, 0);
        pos_embed = ps.add(prefix + "pos_embed", GGML_TYPE_F32, {cfg.hidden_size, cfg.pos_embed_max_size * cfg.pos_embed_max_size, 1}, InitKind::BIAS, cfg.hidden_size);
        t_mlp0.init(ps, prefix + "t_embedder.mlp.0.", 256, cfg.hidden_size, true, false, true);
        t_mlp2.init(ps, prefix + "t_embedder.mlp.2.", cfg.hidden_size, cfg.hidden_size, true, false, true);
        y_mlp0.init(ps, prefix + "y_embedder.mlp.0.", cfg.adm_in_channels, cfg.hidden_size, true, false, true);
        y_mlp2.init(ps, prefix + "y_embedder.mlp.2.", cfg.hidden_size, cfg.hidden_size, true, false, true);
        context_embedder.init(ps, prefix + "context_embedder.", cfg.context_size, cfg.hidden_size, true, false, true);
        blocks.resize(cfg.depth);
        const int64_t heads = cfg.num_heads > 0 ? cfg.num_heads : cfg.depth;
        for (int64_t i = 0; i < cfg.depth; ++i) {  // heads = depth (mmdit.hpp:799), qkv_bias true, last context block pre_only
            const std::string p = prefix + "joint_blocks." + std::to_string(i) + ".";
            blocks[i].context_block.init(ps, p + "context_block.", cfg.hidden_size, heads, cfg.mlp_ratio, cfg.qk_rms, i == cfg.depth - 1, false);
            blocks[i].x_block.init(ps, p + "x_block.", cfg.hidden_size, heads, cfg.mlp_ratio, cfg.qk_rms, false, i <= cfg.d_self);
        }
        final_linear.init(ps, prefix + "final_layer.linear.", cfg.hidden_size, cfg.patch_size * cfg.patch_size * cfg.out_channels, true, false, true);
        final_adaLN.init(ps, prefix + "final_layer.adaLN_modulation.1.", cfg.hidden_size, 2 * cfg.hidden_size);
    }

    // cropped_pos_embed — mmdit.hpp:808-847
    ggml_tensor* cropped_pos_embed(ggml_context* c, int64_t h, int64_t w) const {
        h = (h + 1) / cfg.patch_size;
        w = (w + 1) / cfg.patch_size;
        const int64_t top = (cfg.pos_embed_max_size - h) / 2, left = (cfg.pos_embed_max_size - w) / 2;
        ggml_tensor* sp = ggml_reshape_3d(c, pos_embed, cfg.hidden_size, cfg.pos_embed_max_size, cfg.pos_embed_max_size);
        sp = ggml_view_3d(c, sp, cfg.hidden_size, cfg.pos_embed_max_size, h, sp->nb[1], sp->nb[2], sp->nb[2] * top);
        sp = ggml_cont(c, ggml_permute(c, sp, 0, 2, 1, 3));
        sp = ggml_view_3d(c, sp, cfg.hidden_size, h, w, sp->nb[1], sp->nb[2], sp->nb[2] * left);
        sp = ggml_cont(c, ggml_permute(c, sp, 0, 2, 1, 3));
        return ggml_reshape_3d(c, sp, cfg.hidden_size, h * w, 1);
    }

    // block_mixing — mmdit.hpp:614-699
    void block_mixing(GraphCtx& g, const Joint& jb, ggml_tensor*& context, ggml_tensor*& x, ggml_tensor* cvec) const {
        ggml_context* c = g.ctx;
        auto cp         = jb.context_block.pre_attention(g, context, cvec);
        auto xp         = jb.x_block.pre_attention(g, x, cvec);
        std::vector<ggml_tensor*> qkv;
        for (int i = 0; i < 3; ++i) qkv.push_back(ggml_concat(c, cp.qkv[i], xp.qkv[i], 1));
        ggml_tensor* attn = ext_attention(g, qkv[0], qkv[1], qkv[2], jb.x_block.num_heads);  // [hidden, n_context + n_token, N]
        ggml_tensor* context_attn = ggml_view_3d(c, attn, attn->ne[0], context->ne[1], attn->ne[2], attn->nb[1], attn->nb[2], 0);
        ggml_tensor* x_attn       = ggml_view_3d(c, attn, attn->ne[0], x->ne[1], attn->ne[2], attn->nb[1], attn->nb[2], context->ne[1] * attn->nb[1]);
        ggml_tensor* new_context  = jb.context_block.pre_only ? nullptr : jb.context_block.post_attention(g, context_attn, nullptr, cp.inter);
        if (jb.x_block.self_attn) {
            ggml_tensor* attn2 = ext_attention(g, xp.qkv2[0], xp.qkv2[1], xp.qkv2[2], jb.x_block.num_heads);
            x                  = jb.x_block.post_attention(g, x_attn, attn2, xp.inter);
        } else {
            x = jb.x_block.post_attention(g, x_attn, nullptr, xp.inter);
        }
        context = new_context;
    }

    // forward — mmdit.hpp:881-927.  x [W,H,C,N]; timesteps [N]; context [context_size, L, N|1|2]; y [adm, N|1|2] -> [W,H,C,N]
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* timesteps, ggml_tensor* context, ggml_tensor* y) const {
        ggml_context* c = g.ctx;
        const int64_t W = x->ne[0], H = x->ne[1], N = x->ne[3];
        // OUR extension (batch > 1 per graph): conditioning given once (or once per cfg branch) is tiled over the images
        if (context != nullptr && context->ne[2] != N) context = ggml_repeat(c, context, ggml_new_tensor_3d(c, GGML_TYPE_F32, context->ne[0], context->ne[1], N));
        if (y != nullptr && y->ne[1] != N) y = ggml_repeat(c, y, ggml_new_tensor_2d(c, GGML_TYPE_F32, y->ne[0], N));

        // PatchEmbed (mmdit.hpp:216-243): pad to the patch grid, stride-p conv, flatten to tokens
        const int ps_ = cfg.patch_size;
        const int pad_h = (ps_ - (int)(H % ps_)) % ps_, pad_w = (ps_ - (int)(W % ps_)) % ps_;
        ggml_tensor* pe = ggml_pad(c, x, pad_w, pad_h, 0, 0);
        pe              = x_embedder.forward(g, pe);
        pe              = ggml_reshape_3d(c, pe, pe->ne[0] * pe->ne[1], pe->ne[2], pe->ne[3]);
        pe              = ggml_cont(c, ggml_permute(c, pe, 1, 0, 2, 3));  // [hidden, h*w, N]
        x               = ggml_add(c, pe, cropped_pos_embed(c, H, W));

        ggml_tensor* cv = ggml_timestep_embedding(c, ext_scale(c, timesteps, 1.0f), 256, 10000);  // TimestepEmbedder, mmdit.hpp:245-272
        cv              = t_mlp0.forward(g, cv);
        cv              = ggml_silu_inplace(c, cv);
        cv              = t_mlp2.forward(g, cv);
        if (y != nullptr) {  // VectorEmbedder, mmdit.hpp:274-297
            ggml_tensor* ye = y_mlp0.forward(g, y);
            ye              = ggml_silu_inplace(c, ye);
            ye              = y_mlp2.forward(g, ye);
            cv              = ggml_add(c, cv, ye);
        }
        if (context != nullptr) context = context_embedder.forward(g, context);
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            // skip-layer guidance (mmdit.hpp:862-865): the listed joint blocks are left out of this forward
            if (g.skip_layers && std::find(g.skip_layers->begin(), g.skip_layers->end(), (int)bi) != g.skip_layers->end()) continue;
            block_mixing(g, blocks[bi], context, x, cv);
        }

        // FinalLayer — mmdit.hpp:725-757
        auto mv = ext_chunk(c, final_adaLN.forward(g, ggml_silu(c, cv)), 2, 0, true);
        x       = modulate(c, norm_final.forward(g, x), mv[0], mv[1]);
        x       = final_linear.forward(g, x);

        // DiT::unpatchify_and_crop(patch_last = false) — dit.hpp:36-104
        const int64_t h = (H + pad_h) / ps_, w = (W + pad_w) / ps_, C = cfg.out_channels;
        x = ggml_reshape_4d(c, x, C, ps_ * ps_, w * h, N);
        x = ggml_cont(c, ggml_permute(c, x, 2, 0, 1, 3));  // [ph*pw, h*w, C, N]
        x = ggml_reshape_4d(c, x, ps_, ps_, w, h * C * N);
        x = ggml_cont(c, ggml_permute(c, x, 0, 2, 1, 3));
        x = ggml_reshape_4d(c, x, w * ps_, h * ps_, C, N);
        x = ext_slice(c, x, 1, 0, H);
        x = ext_slice(c, x, 0, 0, W);
        return x;
    }
};

// =====================================================================================================
// FLUX.1 — src/model/diffusion/flux.hpp (FLUX.1-dev: guidance-distilled, 19 double + 38 single stream blocks)
// =====================================================================================================
struct FluxConfig {  // flux.hpp:28-60
    int patch_size            = 2;
    int64_t in_channels       = 64;  // 16 latent channels x 2x2 patch
    int64_t out_channels      = 64;
    int64_t vec_in_dim        = 768;
    int64_t context_in_dim    = 4096;
    int64_t hidden_size       = 3072;
    float mlp_ratio           = 4.0f;
    int num_heads             = 24;
    int depth                 = 19;
    int depth_single_blocks   = 38;
    std::vector<int> axes_dim = {16, 56, 56};
    int theta                 = 10000;
    bool guidance_embed       = true;
    static FluxConfig flux_dev() { return FluxConfig(); }
    // FLUX.1-dev's real width (hidden 3072, 24 heads x 128, RoPE 16/56/56) with ONE double and ONE single stream block
    static FluxConfig flux_wide1() {
        FluxConfig c;
        c.depth               = 1;
        c.depth_single_blocks = 1;
        return c;
    }
    static FluxConfig flux_wide8() {  // 3 double + 5 single blocks at the real width (depth sweep of the full-depth parity test)
        FluxConfig c;
        c.depth               = 3;
        c.depth_single_blocks = 5;
        return c;
    }
    static FluxConfig tiny() {
        FluxConfig c;
        c.vec_in_dim          = 64;
        c.context_in_dim      = 96;
        c.hidden_size         = 128;
        c.num_heads           = 4;  // d_head 32 = 8 + 12 + 12
        c.depth               = 2;
        c.depth_single_blocks = 2;
        c.axes_dim            = {8, 12, 12};
        return c;
    }
};

// Rope::gen_flux_pe (rope.hpp:55-106, 130-250, 398-490): ids = (0, row, col) for image patches, 0 for text tokens (text first);
// per axis the rotation angles pos * theta^(-2j/dim) as 2x2 matrices [[cos, -sin], [sin, cos]] -> [L][d_head/2][2][2] floats
inline std::vector<float> gen_flux_pe(int h, int w, int patch_size, int context_len, const std::vector<int>& axes_dim, float theta) {
    const int h_len = (h + patch_size / 2) / patch_size, w_len = (w + patch_size / 2) / patch_size;
    const int L = context_len + h_len * w_len;
    int half_sum = 0;
    for (int d : axes_dim) half_sum += d / 2;
    std::vector<float> pe((size_t)L * half_sum * 4, 0.f);
    for (int pos = 0; pos < L; ++pos) {
        float ids[3] = {0.f, 0.f, 0.f};
        if (pos >= context_len) {
            const int p = pos - context_len;
            ids[1]      = (float)(p / w_len);
            ids[2]      = (float)(p % w_len);
        }
        size_t off = (size_t)pos * half_sum * 4;
        for (size_t a = 0; a < axes_dim.size(); ++a) {
            const int dim = axes_dim[a], half = dim / 2;
            for (int j = 0; j < half; ++j) {
                // linspace(0, (dim-2)/dim, half)[j]
                const float sc    = half == 1 ? 0.f : ((dim * 1.f - 2) / dim) / (half - 1) * j;
                const float omega = 1.0f / ::powf(theta, sc);
                const float ang   = ids[a] * omega;
                pe[off + 4 * j]     = std::cos(ang);
                pe[off + 4 * j + 1] = -std::sin(ang);
                pe[off + 4 * j + 2] = std::sin(ang);
                pe[off + 4 * j + 3] = std::cos(ang);
            }
            off += (size_t)half * 4;
        }
    }
    return pe;
}

struct FluxMLPEmbedder {  // flux.hpp:193-211; time_in / vector_in / guidance_in are never quantised (model_loader.cpp:1523-1529)
    Linear in_layer, out_layer;
    void init(ParamStore& ps, const std::string& prefix, int64_t in, int64_t hidden) {
        in_layer.init(ps, prefix + "in_layer.", in, hidden, true, true);
        out_layer.init(ps, prefix + "out_layer.", hidden, hidden, true, true);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const {
        x = in_layer.forward(g, x);
        x = ggml_silu_inplace(g.ctx, x);
        return out_layer.forward(g, x);
    }
};

struct FluxRMSNorm {  // flux.hpp:213-235: parameter is called "scale", the multiply is NOT in place
    ggml_tensor* w = nullptr;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim) { w = ps.add(prefix + "scale", GGML_TYPE_F32, {dim}, InitKind::NORM_SCALE, dim); }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return ggml_mul(g.ctx, ggml_rms_norm(g.ctx, x, 1e-6f), w); }
};

struct FluxModOut {
    ggml_tensor *shift = nullptr, *scale = nullptr, *gate = nullptr;
};
struct FluxModulation {  // flux.hpp:381-411
    bool is_double = false;
    Linear lin;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, bool dbl) {
        is_double = dbl;
        lin.init(ps, prefix + "lin.", dim, dim * (dbl ? 6 : 3));
    }
    std::vector<FluxModOut> forward(GraphCtx& g, ggml_tensor* vec) const {
        ggml_context* c = g.ctx;
        const int mult  = is_double ? 6 : 3;
        ggml_tensor* m  = lin.forward(g, ggml_silu(c, vec));
        m               = ggml_reshape_3d(c, m, vec->ne[0], mult, vec->ne[1]);
        m               = ggml_cont(c, ggml_permute(c, m, 0, 2, 1, 3));  // [dim, N, mult]
        const size_t st = m->nb[1] * m->ne[1];
        auto out        = [&](int o) {
            FluxModOut r;
            r.shift = ggml_view_2d(c, m, m->ne[0], m->ne[1], m->nb[1], st * (o + 0));
            r.scale = ggml_view_2d(c, m, m->ne[0], m->ne[1], m->nb[1], st * (o + 1));
            r.gate  = ggml_view_2d(c, m, m->ne[0], m->ne[1], m->nb[1], st * (o + 2));
            return r;
        };
        if (is_double) return {out(0), out(3)};
        return {out(0), FluxModOut()};
    }
};

struct FluxSelfAttention {  // flux.hpp:263-315
    int64_t num_heads = 0;
    Linear qkv, proj;
    FluxRMSNorm query_norm, key_norm;
    void init(ParamStore& ps, const std::string& prefix, int64_t dim, int64_t heads) {
        num_heads = heads;
        qkv.init(ps, prefix + "qkv.", dim, dim * 3, true);
        query_norm.init(ps, prefix + "norm.query_norm.", dim / heads);
        key_norm.init(ps, prefix + "norm.key_norm.", dim / heads);
        proj.init(ps, prefix + "proj.", dim, dim, true);
    }
    std::vector<ggml_tensor*> pre_attention(GraphCtx& g, ggml_tensor* x) const {
        ggml_context* c   = g.ctx;
        ggml_tensor* t    = qkv.forward(g, x);
        const int64_t hd  = t->ne[0] / 3 / num_heads;
        auto part         = [&](int i) { return ggml_view_4d(c, t, hd, num_heads, t->ne[1], t->ne[2], t->nb[0] * hd, t->nb[1], t->nb[2], t->nb[0] * t->ne[0] / 3 * i); };
        return {query_norm.forward(g, part(0)), key_norm.forward(g, part(1)), part(2)};
    }
};

struct FluxMLP {  // flux.hpp:317-341 (GELU tanh)
    Linear l0, l2;
    void init(ParamStore& ps, const std::string& prefix, int64_t hidden, int64_t inter) {
        l0.init(ps, prefix + "0.", hidden, inter);
        l2.init(ps, prefix + "2.", inter, hidden);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x) const { return l2.forward(g, ext_gelu(g.ctx, l0.forward(g, x), true)); }
};

// gate operand of `x + out * gate` in the FLUX blocks: the [H, N] view itself for N == 1 (what the reference emits), [H, 1, N] for our batched graphs
inline ggml_tensor* flux_gate(ggml_context* c, ggml_tensor* gate) { return gate->ne[1] == 1 ? gate : ggml_reshape_3d(c, gate, gate->ne[0], 1, gate->ne[1]); }

struct FluxDoubleBlock {  // flux.hpp:430-592
    FluxModulation img_mod, txt_mod;
    PlainLayerNorm n1, n2;
    FluxSelfAttention img_attn, txt_attn;
    FluxMLP img_mlp, txt_mlp;
    void init(ParamStore& ps, const std::string& prefix, const FluxConfig& cfg) {
        const int64_t H = cfg.hidden_size, mh = (int64_t)(H * cfg.mlp_ratio);
        img_mod.init(ps, prefix + "img_mod.", H, true);
        img_attn.init(ps, prefix + "img_attn.", H, cfg.num_heads);
        img_mlp.init(ps, prefix + "img_mlp.", H, mh);
        txt_mod.init(ps, prefix + "txt_mod.", H, true);
        txt_attn.init(ps, prefix + "txt_attn.", H, cfg.num_heads);
        txt_mlp.init(ps, prefix + "txt_mlp.", H, mh);
    }
    void forward(GraphCtx& g, ggml_tensor*& img, ggml_tensor*& txt, ggml_tensor* vec, ggml_tensor* pe) const {
        ggml_context* c = g.ctx;
        auto im = img_mod.forward(g, vec), tm = txt_mod.forward(g, vec);
        auto iq = img_attn.pre_attention(g, modulate(c, n1.forward(g, img), im[0].shift, im[0].scale));
        auto tq = txt_attn.pre_attention(g, modulate(c, n1.forward(g, txt), tm[0].shift, tm[0].scale));
        ggml_tensor* q = ggml_concat(c, tq[0], iq[0], 2);
        ggml_tensor* k = ggml_concat(c, tq[1], iq[1], 2);
        ggml_tensor* v = ggml_concat(c, tq[2], iq[2], 2);
        ggml_tensor* attn = rope_attention(g, q, k, v, pe);  // [H, n_txt + n_img, N]
        ggml_tensor* txt_attn_out = ggml_view_3d(c, attn, attn->ne[0], txt->ne[1], attn->ne[2], attn->nb[1], attn->nb[2], 0);
        ggml_tensor* img_attn_out = ggml_view_3d(c, attn, attn->ne[0], img->ne[1], attn->ne[2], attn->nb[1], attn->nb[2], txt->ne[1] * attn->nb[1]);
        // the reference multiplies by the [H, N] gate view directly (flux.hpp:578-588), which only broadcasts for N == 1 (flux.hpp:1281 asserts it): for
        // N == 1 the graph is the reference's node for node (tests/test_ref_graphs.py — until round 6 a RESHAPE sat in front of every gate MUL);
        // OUR batched graphs (N > 1) need the [H, 1, N] form
        img = ggml_add(c, img, ggml_mul(c, img_attn.proj.forward(g, img_attn_out), flux_gate(c, im[0].gate)));
        ggml_tensor* imlp = img_mlp.forward(g, modulate(c, n2.forward(g, img), im[1].shift, im[1].scale));
        img = ggml_add(c, img, ggml_mul(c, imlp, flux_gate(c, im[1].gate)));
        txt = ggml_add(c, txt, ggml_mul(c, txt_attn.proj.forward(g, txt_attn_out), flux_gate(c, tm[0].gate)));
        ggml_tensor* tmlp = txt_mlp.forward(g, modulate(c, n2.forward(g, txt), tm[1].shift, tm[1].scale));
        txt = ggml_add(c, txt, ggml_mul(c, tmlp, flux_gate(c, tm[1].gate)));
    }
};

struct FluxSingleBlock {  // flux.hpp:594-700
    int64_t hidden = 0, heads = 0, mlp_hidden = 0;
    Linear linear1, linear2;
    FluxRMSNorm query_norm, key_norm;
    PlainLayerNorm pre_norm;
    FluxModulation modulation;
    void init(ParamStore& ps, const std::string& prefix, const FluxConfig& cfg) {
        hidden     = cfg.hidden_size;
        heads      = cfg.num_heads;
        mlp_hidden = (int64_t)(hidden * cfg.mlp_ratio);
        linear1.init(ps, prefix + "linear1.", hidden, hidden * 3 + mlp_hidden);
        linear2.init(ps, prefix + "linear2.", hidden + mlp_hidden, hidden);
        query_norm.init(ps, prefix + "norm.query_norm.", hidden / heads);
        key_norm.init(ps, prefix + "norm.key_norm.", hidden / heads);
        modulation.init(ps, prefix + "modulation.", hidden, false);
    }
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* vec, ggml_tensor* pe) const {
        ggml_context* c  = g.ctx;
        FluxModOut mod   = modulation.forward(g, vec)[0];
        ggml_tensor* t   = linear1.forward(g, modulate(c, pre_norm.forward(g, x), mod.shift, mod.scale));
        const int64_t hd = hidden / heads;
        auto part        = [&](int i) { return ggml_view_4d(c, t, hd, heads, t->ne[1], t->ne[2], t->nb[0] * hd, t->nb[1], t->nb[2], t->nb[0] * hidden * i); };
        ggml_tensor* attn = rope_attention(g, query_norm.forward(g, part(0)), key_norm.forward(g, part(1)), part(2), pe);
        ggml_tensor* mlp  = ggml_view_3d(c, t, mlp_hidden, t->ne[1], t->ne[2], t->nb[1], t->nb[2], hidden * 3 * t->nb[0]);
        mlp               = ext_gelu(c, mlp, true);
        ggml_tensor* out  = linear2.forward(g, ggml_concat(c, attn, mlp, 0));
        return ggml_add(c, x, ggml_mul(c, out, flux_gate(c, mod.gate)));
    }
};

struct FluxModel {
    FluxConfig cfg;
    Linear img_in, txt_in, final_linear, final_adaLN;
    FluxMLPEmbedder time_in, vector_in, guidance_in;
    PlainLayerNorm norm_final;
    std::vector<FluxDoubleBlock> double_blocks;
    std::vector<FluxSingleBlock> single_blocks;

    void init(ParamStore& ps, const std::string& prefix, const FluxConfig& c) {
        cfg = c;
        // img_in / txt_in / final_layer keep f16 under a quantised wtype (tensor_should_be_converted, model_loader.cpp:1523-1529)
        img_in.init(ps, prefix + "img_in.", cfg.in_channels, cfg.hidden_size, true, true);
        time_in.init(ps, prefix + "time_in.", 256, cfg.hidden_size);
        vector_in.init(ps, prefix + "vector_in.", cfg.vec_in_dim, cfg.hidden_size);
        if (cfg.guidance_embed) guidance_in.init(ps, prefix + "guidance_in.", 256, cfg.hidden_size);
        txt_in.init(ps, prefix + "txt_in.", cfg.context_in_dim, cfg.hidden_size, true, true);
        double_blocks.resize(cfg.depth);
        for (int i = 0; i < cfg.depth; ++i) double_blocks[i].init(ps, prefix + "double_blocks." + std::to_string(i) + ".", cfg);
        single_blocks.resize(cfg.depth_single_blocks);
        for (int i = 0; i < cfg.depth_single_blocks; ++i) single_blocks[i].init(ps, prefix + "single_blocks." + std::to_string(i) + ".", cfg);
        final_linear.init(ps, prefix + "final_layer.linear.", cfg.hidden_size, cfg.out_channels, true, true);
        final_adaLN.init(ps, prefix + "final_layer.adaLN_modulation.1.", cfg.hidden_size, 2 * cfg.hidden_size, true, true);
    }

    // forward_flux_chroma + forward_orig — flux.hpp:1267-1337, 1008-1182.  x [W,H,16,N]; timestep [N] (= sigma); context [ctx, L, N|1|2];
    // y [vec, N|1|2]; guidance [N]; pe [2,2,d_head/2, L_txt + L_img] (host-built, gen_flux_pe) -> [W,H,16,N]
    ggml_tensor* forward(GraphCtx& g, ggml_tensor* x, ggml_tensor* timestep, ggml_tensor* context, ggml_tensor* y, ggml_tensor* guidance, ggml_tensor* pe) const {
        ggml_context* c = g.ctx;
        const int64_t W = x->ne[0], H = x->ne[1], C = x->ne[2], N = x->ne[3];
        // OUR extension (batch > 1 per graph; the reference asserts N == 1, flux.hpp:1281): conditioning is tiled over the images
        if (context->ne[2] != N) context = ggml_repeat(c, context, ggml_new_tensor_3d(c, GGML_TYPE_F32, context->ne[0], context->ne[1], N));
        if (y->ne[1] != N) y = ggml_repeat(c, y, ggml_new_tensor_2d(c, GGML_TYPE_F32, y->ne[0], N));
        const int ps_ = cfg.patch_size;
        const int pad_h = (ps_ - (int)(H % ps_)) % ps_, pad_w = (ps_ - (int)(W % ps_)) % ps_;
        // DiT::pad_and_patchify(patch_last = true) — dit.hpp:7-34, 67-88; ggml_ext_pad emits the PAD node only when there is something to pad
        // (ggml_extend.hpp:1100-1113; round 6: found against the reference-emitted graph)
        ggml_tensor* img = (pad_w != 0 || pad_h != 0) ? ggml_pad(c, x, pad_w, pad_h, 0, 0) : x;
        const int64_t h = (H + pad_h) / ps_, w = (W + pad_w) / ps_;
        img = ggml_reshape_4d(c, img, ps_, w, ps_, h * C * N);
        img = ggml_cont(c, ggml_permute(c, img, 0, 2, 1, 3));
        img = ggml_reshape_4d(c, img, ps_ * ps_, w * h, C, N);
        img = ggml_cont(c, ggml_permute(c, img, 0, 2, 1, 3));
        img = ggml_reshape_3d(c, img, ps_ * ps_ * C, w * h, N);

        img = img_in.forward(g, img);
        ggml_tensor* vec = time_in.forward(g, ggml_timestep_embedding(c, ext_scale(c, timestep, 1000.f), 256, 10000));
        if (cfg.guidance_embed) vec = ggml_add(c, vec, guidance_in.forward(g, ggml_timestep_embedding(c, ext_scale(c, guidance, 1000.f), 256, 10000)));
        vec              = ggml_add(c, vec, vector_in.forward(g, y));
        ggml_tensor* txt = txt_in.forward(g, context);
        for (auto& b : double_blocks) b.forward(g, img, txt, vec, pe);
        ggml_tensor* txt_img = ggml_concat(c, txt, img, 1);
        for (auto& b : single_blocks) txt_img = b.forward(g, txt_img, vec, pe);
        img = ggml_view_3d(c, txt_img, txt_img->ne[0], img->ne[1], txt_img->ne[2], txt_img->nb[1], txt_img->nb[2], txt->ne[1] * txt_img->nb[1]);
        // LastLayer — flux.hpp:702-757
        auto mv = ext_chunk(c, final_adaLN.forward(g, ggml_silu(c, vec)), 2, 0, true);
        img     = modulate(c, norm_final.forward(g, img), mv[0], mv[1]);
        img     = final_linear.forward(g, img);  // [ps*ps*C, h*w, N], patch last
        // DiT::unpatchify_and_crop(patch_last = true) — dit.hpp:36-104
        img = ggml_reshape_4d(c, img, ps_ * ps_, C, w * h, N);
        img = ggml_cont(c, ggml_permute(c, img, 0, 2, 1, 3));
        img = ggml_reshape_4d(c, img, ps_, ps_, w, h * C * N);
        img = ggml_cont(c, ggml_permute(c, img, 0, 2, 1, 3));
        img = ggml_reshape_4d(c, img, w * ps_, h * ps_, C, N);
        img = ext_slice(c, img, 1, 0, H);
        img = ext_slice(c, img, 0, 0, W);
        return img;
    }
};

}  // namespace sdmi
