/*
 * sd-mi355x.h — C ABI of the MI355X denoise + VAE-decode engine (libsdcpp-host.so).
 *
 * This is the host-facing slice of the reference's public C API (include/stable-diffusion.h) that the
 * hot path touches, with the same names / argument meaning / error behaviour where a counterpart
 * exists, and plain pointers + sizes everywhere (no C++ / torch types).  What differs, and why:
 *   - conditioning enters the denoise path as tensors (`sd_condition_t`, the SDCondition struct of
 *     src/conditioning/conditioner.hpp:18-34); the text encoders that produce them take token ids
 *     (sd_get_learned_condition below) because the reference's tokenizer vocabularies are not in its source drop;
 *   - weights are synthetic (random-init of the named architecture) unless the caller overwrites
 *     tensors by name with sd_set_tensor (checkpoint readers are a "next" row).
 *
 * All compute goes through ggml's backend plug-in interface (include/ggml-abi.h).  The engine loads
 * libggml-mi355x.so and FAILS (returns NULL) if it or a gfx950 device is missing — there is no CPU
 * fallback in the product.  Tests may register another backend plug-in (the CPU oracle) explicitly
 * through sd_load_backend() and select it by name.
 *
 * NAMING.  The reference's public API takes model paths and prompt strings and its parameter structs carry ~50 fields the hot path never
 * reads (include/stable-diffusion.h:192-287); this slice takes tensors and token ids, so its structs CANNOT be layout-compatible.  Every
 * type, enumerator and function whose name exists in the reference header therefore carries the prefix `sdm_` / `SDM_` here: a program
 * may include both headers and link both libraries, and a reference-side caller can never bind one of these entry points by accident
 * (round-1 review: same names + different layouts = silent memory corruption).  The pairs:
 *
 * reference (stable-diffusion.h)                  -> this header
 *   sd_ctx_params_t                     :192-238 -> sdm_ctx_params_t (only the fields the hot path reads; + model family, weight seed)
 *   sd_sample_params_t                  :276-287 -> sdm_sample_params_t
 *   sd_img_gen_params_t                          -> sdm_img_gen_params_t (conditioning as tensors instead of prompt strings)
 *   sd_image_t                          :258-263 -> sdm_image_t (same four fields, same order)
 *   enum sample_method_t / scheduler_t  :38-83   -> sdm_sample_method_t / sdm_scheduler_t (same numeric values for the members kept)
 *   enum sd_type_t                      :99-143  -> sdm_type_t (same numeric values = enum ggml_type)
 *   new_sd_ctx / free_sd_ctx            :482-485 -> sdm_new_ctx / sdm_free_ctx
 *   generate_image                      :493-496 -> sdm_generate_image (batch_count images, seeds seed+b)
 *   free_sd_images                      :595-597 -> sdm_free_images
 *   sd_*_params_init                             -> sdm_*_params_init
 *   sd_set_backend_eval_callback        :442-447 -> sdm_set_backend_eval_callback (sub-graph views are honoured by the backend)
 *   sd_img_gen_params_t::init_image / mask_image / strength -> sdm_img_gen_params_t::init_latent (sd_vae_encode) / denoise_mask / strength
 *   sd_sample_params_t::custom_sigmas / flow_shift / guidance.slg -> the same fields of sdm_sample_params_t (slg_*)
 *   sd_ctx_params_t::prediction / taesd_path     -> sd_set_prediction / sd_use_tae + sd_load_weights_prefixed(ctx, file, "tae.")
 * Names without a reference counterpart (sd_unet_forward, sd_vae_decode, sd_load_backend, ...) keep the plain `sd_` prefix.
 */
#ifndef SD_MI355X_H
#define SD_MI355X_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_API __attribute__((visibility("default")))

enum sd_model_family_t {
    SD_MODEL_SD15       = 0, /* UNetConfig defaults, unet.hpp:16-35; VAE 4ch scale 0.18215 */
    SD_MODEL_SDXL       = 1, /* unet.hpp:47-57; VAE scale 0.13025 */
    SD_MODEL_SD15_TINY  = 2, /* same topology, model_channels 32 — CPU-sized parity tests */
    SD_MODEL_SDXL_TINY  = 3,
    SD_MODEL_SD35_LARGE = 4, /* MMDiT, 38 joint blocks, hidden 2432, rms qk-norm (mmdit.hpp); 16-ch VAE scale 1.5305 shift 0.0609;
                                discrete-flow denoiser, shift 3.0 (denoiser.hpp:1239-1283) */
    SD_MODEL_SD35_TINY  = 5, /* same topology at 3 blocks / hidden 192, with one MMDiT-X self-attention block */
    SD_MODEL_FLUX_DEV   = 6, /* FLUX.1-dev: 19 double + 38 single stream blocks, hidden 3072, 24 heads, RoPE axes 16/56/56, distilled
                                guidance input (flux.hpp); 16-ch VAE scale 0.3611 shift 0.1159; FluxFlowDenoiser + Flux scheduler */
    SD_MODEL_FLUX_TINY  = 7, /* same topology: 2 + 2 blocks, hidden 128, 4 heads, axes 8/12/12 */
    SD_MODEL_SD35_WIDE2 = 8, /* SD3.5-large's real width (hidden 2432, 38 heads x 64), 2 joint blocks — full-width block parity tests */
    SD_MODEL_FLUX_WIDE1 = 9, /* FLUX.1-dev's real width (hidden 3072, 24 heads x 128), 1 double + 1 single block — same purpose */
    SD_MODEL_SD3M_TINY  = 10, /* SD3-medium's topology at test size: MMDiT WITHOUT qk-norm and without MMDiT-X blocks (mmdit.hpp:299-366 with qk_norm = "") */
    SD_MODEL_SD35_WIDE8 = 11, /* SD3.5-large's real width, 8 joint blocks: the middle point of the depth sweep (2 / 8 / 38) of tests/test_zz_gpu_fulldepth.py */
    SD_MODEL_FLUX_WIDE8 = 12, /* FLUX.1-dev's real width, 3 double + 5 single blocks: the middle point of the FLUX depth sweep (1+1 / 3+5 / 19+38) */
};

/* numeric values = enum ggml_type (stable-diffusion.h:98-143) */
enum sdm_type_t {
    SDM_TYPE_F32  = 0,
    SDM_TYPE_F16  = 1,
    SDM_TYPE_Q4_0 = 2,
    SDM_TYPE_Q8_0 = 8,
    SDM_TYPE_BF16 = 30,
};

/* SDM_SAMPLE_METHOD_COUNT = "the family's default" like the reference's SAMPLE_METHOD_COUNT (sd_get_default_sample_method,
 * stable-diffusion.cpp:3965-3975): Euler for the DiT families (SD3.5, FLUX), Euler-A otherwise */
/* Numeric values = the reference's sample_method_t / scheduler_t (include/stable-diffusion.h:38-83), so integers a host already holds keep their meaning.  Implemented
 * (host-side math, bit-for-bit against src/runtime/denoiser.hpp compiled from the reference: tests/test_host_logic.py): EVERY method sample_k_diffusion dispatches, 0 ... 20 (the `extra_sample_args` knobs of LMS / Euler GE / LCM
 * at their defaults); schedulers DISCRETE, KARRAS, EXPONENTIAL, AYS, GITS, SGM_UNIFORM, SIMPLE, SMOOTHSTEP, KL_OPTIMAL, LCM, BONG_TANGENT, BETA (alpha = beta = 0.6) and FLUX — every
 * scheduler_t value but the default ladders of model families outside this engine (LTX2 = 11, LOGIT_NORMAL = 12, FLUX2 = 13).  Any other value makes sdm_generate_image / sdm_sample_latents fail with an error message
 * (sd_last_error) instead of silently sampling something else.  SDM_SCHEDULER_COUNT = "the default for this model and method" (sd_get_default_scheduler,
 * stable-diffusion.cpp:3977-3998): LCM for the LCM and TCD methods, SIMPLE for DDIM trailing, FLUX for FLUX, DISCRETE otherwise.  Euler / Euler-A (and DDIM trailing, which the
 * reference runs as Euler-A) take the device-resident sampler; the multi-stage and multi-step methods run the host loop around the device forward. */
enum sdm_sample_method_t {
    SDM_EULER_SAMPLE_METHOD = 0, SDM_EULER_A_SAMPLE_METHOD = 1, SDM_HEUN_SAMPLE_METHOD = 2, SDM_DPM2_SAMPLE_METHOD = 3, SDM_DPMPP2S_A_SAMPLE_METHOD = 4,
    SDM_DPMPP2M_SAMPLE_METHOD = 5, SDM_DPMPP2Mv2_SAMPLE_METHOD = 6, SDM_IPNDM_SAMPLE_METHOD = 7, SDM_IPNDM_V_SAMPLE_METHOD = 8, SDM_LCM_SAMPLE_METHOD = 9,
    SDM_DDIM_TRAILING_SAMPLE_METHOD = 10, SDM_TCD_SAMPLE_METHOD = 11, SDM_RES_MULTISTEP_SAMPLE_METHOD = 12, SDM_RES_2S_SAMPLE_METHOD = 13, SDM_ER_SDE_SAMPLE_METHOD = 14,
    SDM_EULER_CFG_PP_SAMPLE_METHOD = 15, SDM_EULER_A_CFG_PP_SAMPLE_METHOD = 16, SDM_EULER_GE_SAMPLE_METHOD = 17, SDM_DPMPP2M_SDE_SAMPLE_METHOD = 18,
    SDM_DPMPP2M_SDE_BT_SAMPLE_METHOD = 19, SDM_LMS_SAMPLE_METHOD = 20, SDM_SAMPLE_METHOD_COUNT = 21
};
enum sdm_scheduler_t {
    SDM_DISCRETE_SCHEDULER = 0, SDM_KARRAS_SCHEDULER = 1, SDM_EXPONENTIAL_SCHEDULER = 2, SDM_AYS_SCHEDULER = 3, SDM_GITS_SCHEDULER = 4, SDM_SGM_UNIFORM_SCHEDULER = 5,
    SDM_SIMPLE_SCHEDULER = 6, SDM_SMOOTHSTEP_SCHEDULER = 7, SDM_KL_OPTIMAL_SCHEDULER = 8, SDM_LCM_SCHEDULER = 9, SDM_BONG_TANGENT_SCHEDULER = 10, SDM_FLUX_SCHEDULER = 14,
    SDM_BETA_SCHEDULER = 15, SDM_SCHEDULER_COUNT = 16
};

/* prediction_t of the reference (include/stable-diffusion.h), the two UNet parameterisations: sd_set_prediction */
enum sdm_prediction_t { SDM_EPS_PRED = 0, SDM_V_PRED = 1 };

typedef struct {
    const char* backend;        /* ggml device name, case-insensitive; NULL -> "MI355X0" (stable-diffusion.h:232) */
    enum sd_model_family_t model;
    enum sdm_type_t wtype;       /* Linear weight type (conv stays f16, norms/bias f32 — SURVEY.md F9) */
    bool diffusion_flash_attn;  /* stable-diffusion.h:223 */
    bool diffusion_conv_direct; /* stable-diffusion.h:225 */
    bool vae_decode_only;       /* always true here */
    uint64_t weight_seed;       /* synthetic-weight seed (1234) */
    int n_threads;              /* only reaches CPU backends (ggml_extend.hpp:2838-2843) */
} sdm_ctx_params_t;

typedef struct {
    float txt_cfg;              /* cfg scale (7.0 default, stable-diffusion.cpp:3650-3667) */
    enum sdm_scheduler_t scheduler;
    enum sdm_sample_method_t sample_method;
    int sample_steps;
    float eta;                  /* INFINITY -> 1.0 for Euler-A (stable-diffusion.cpp:4024-4049) */
    const float* custom_sigmas; /* sd_sample_params_t::custom_sigmas: count > 1 replaces the scheduler's ladder as it is (last value normally 0) */
    int custom_sigmas_count;
    const int* slg_layers;      /* skip-layer guidance (sd_slg_params_t; SD3.x): joint blocks left out of one more conditional forward per step inside the window */
    int slg_layer_count;        /* (slg_layer_start, slg_layer_end) x the ladder length; guided += (cond - skip) * slg_scale (guidance.cpp:296-340).  scale 0 = off */
    float slg_layer_start, slg_layer_end, slg_scale;
    float apg_eta, apg_momentum, apg_norm_threshold, apg_norm_threshold_smoothing; /* adaptive projected guidance (the reference's extra_sample_args apg_eta / apg_momentum /
                                   apg_norm_threshold / apg_norm_threshold_smoothing; AdaptiveProjectedGuidance, guidance.cpp:209-294) in place of plain CFG on the
                                   (cond, uncond) pair; neutral values (1, 0, 0, 0) = off */
    int shifted_timestep;       /* sd_sample_params_t::shifted_timestep (timestep-shifted distilled UNets): > 0: the model sees round(t * shifted_timestep / 1000) and that
                                   timestep's output scalings (prepare_sample_timesteps / adjust_sample_step_scalings, stable-diffusion.cpp:2411-2457); 0 = off */
    float flow_shift;           /* flow families: the time shift of DiscreteFlowDenoiser (set_flow_shift, stable-diffusion.cpp:3106-3115); INFINITY = the default (SD3.x 3.0, FLUX.1-dev 1.15) */
} sdm_sample_params_t;

/* SDCondition (conditioner.hpp:18-34): c_crossattn [ctx_dim, n_tokens], c_vector [adm] (SDXL) */
typedef struct {
    const float* c_crossattn;
    int64_t ctx_dim, n_tokens;
    const float* c_vector; /* NULL for SD1.x */
    int64_t vector_dim;
} sd_condition_t;

typedef struct {
    sd_condition_t cond;
    sd_condition_t uncond;      /* used iff txt_cfg != 1 (stable-diffusion.cpp:4249) */
    int width, height;          /* pixels; latent = /8 */
    sdm_sample_params_t sample_params;
    int64_t seed;
    int batch_count;            /* images seed .. seed+batch_count-1 (stable-diffusion.cpp:5664-5683) */
    int device_batch;           /* OUR extension: images denoised together per graph; 0 -> batch_count */
    bool decode;                /* run the VAE and return pixels; false -> latents only */
    bool fuse_cfg_pair;         /* OUR extension: cond and uncond of every image run in ONE graph (N = 2*batch, context [.,.,2]
                                   tiled by the graph's own ggml_repeat, unet.hpp:548-552) instead of two computes per step */
    bool device_sampler;        /* OUR extension: the whole iteration — x*c_in, model pair, CFG combine, Euler(-A) update —
                                   is one graph per step on latents that stay in a backend buffer; nothing crosses back to the host until
                                   the last step (the reference's three crossings per model call: stable-diffusion.cpp:2636-2664, 2855-2896) */
    const float* init_latent;   /* img2img (stable-diffusion.h init_image, already encoded: sd_vae_encode): diffusion latents [w/8, h/8, C] shared by the batch like the
                                   reference's one init image; the trajectory starts from noise_scaling(sigma_0, noise, init_latent) (denoiser.hpp:1181-1186, 1274-1279);
                                   NULL = txt2img */
    const float* denoise_mask;  /* inpainting (stable-diffusion.h mask_image, at latent resolution): [w/8, h/8] f32, 1 = repaint, 0 = keep; with init_latent every denoised
                                   prediction becomes denoised * mask + init_latent * (1 - mask) (stable-diffusion.cpp:2888-2890); NULL = none */
    float strength;             /* with init_latent: < 1 keeps the last t_enc + 2 sigmas of the ladder, t_enc = (int)(steps * strength) (stable-diffusion.cpp:4940-4980); 0.75 default */
} sdm_img_gen_params_t;

typedef struct {
    uint32_t width, height, channel;
    uint8_t* data;
} sdm_image_t;

typedef struct sdm_ctx_t sdm_ctx_t;

/* ---- backend discovery (GGML_BACKEND_DL) ---- */
SD_API bool sd_load_backend(const char* path); /* dlopen a libggml-<name>.so and register its devices */
SD_API int sd_device_count(void);
SD_API const char* sd_device_name(int i);
SD_API const char* sd_device_description(int i);

/* ---- context ---- */
SD_API void sdm_ctx_params_init(sdm_ctx_params_t* p);
SD_API void sdm_sample_params_init(sdm_sample_params_t* p);
SD_API void sdm_img_gen_params_init(sdm_img_gen_params_t* p);
SD_API sdm_ctx_t* sdm_new_ctx(const sdm_ctx_params_t* params); /* NULL on failure (no device, alloc failure) */
SD_API void sdm_free_ctx(sdm_ctx_t* ctx);
SD_API const char* sd_last_error(void);

/* ---- weights by name ("model.diffusion_model.<...>", "first_stage_model.<...>") ---- */
SD_API int64_t sd_tensor_count(sdm_ctx_t* ctx);
SD_API const char* sd_tensor_name(sdm_ctx_t* ctx, int64_t i);
/* ne[4], ggml type; returns false if unknown */
SD_API bool sd_tensor_info(sdm_ctx_t* ctx, const char* name, int64_t* ne, int* type, size_t* nbytes);
SD_API bool sd_get_tensor(sdm_ctx_t* ctx, const char* name, void* dst, size_t nbytes);       /* raw bytes (device -> host) */
SD_API bool sd_get_tensor_f32(sdm_ctx_t* ctx, const char* name, float* dst, int64_t nelem);  /* dequantised */
SD_API bool sd_set_tensor_f32(sdm_ctx_t* ctx, const char* name, const float* src, int64_t nelem); /* converts per stored type */

/* ---- checkpoint files, told apart by their first bytes:
 *   safetensors          F32 / F16 / BF16 native; F64 / I64 / F8_E4M3 / F8_E5M2 widened at load
 *   GGUF v2 / v3         F32 / F16 / BF16 / Q8_0 / Q4_0 native; Q4_1 / Q5_0 / Q5_1 / Q2_K / Q3_K / Q4_K / Q5_K / Q6_K / IQ4_NL decoded at load
 *   PyTorch checkpoints  torch.save's zip container (.ckpt / .pt / .pth / .bin, PyTorch >= 1.6) and the legacy stream: float / half / bfloat16 /
 *                        double / long storages of contiguous tensors, found in the root dictionary and the dictionaries nested in it ("state_dict");
 *                        the pickle is interpreted, never executed (csrc/host/torch_ckpt_io.hpp)
 * Every parameter the model declares that the file names (original-LDM / sd.cpp GGUF names, e.g. "model.diffusion_model.input_blocks.0.0.weight")
 * is converted file dtype -> f32 -> the parameter's type (ModelLoader convert_tensor, src/model_loader.cpp:155-205) and uploaded.
 * Returns the number of parameters loaded, -1 on error (sd_last_error); *n_missing = declared but absent, *n_unused = in the file but unknown. */
SD_API int64_t sd_load_weights(sdm_ctx_t* ctx, const char* path, int64_t* n_missing, int64_t* n_unused);
/* Same, with `prefix` prepended to every file name first — diffusers keeps one file per sub-model whose names carry no component prefix
 * ("unet." / "vae." / "text_encoder." / "text_encoder_2.", model_loader.cpp init_from_diffusers_file).  File names are rewritten to the
 * engine's canonical dialect before matching (src/name_conversion.cpp): diffusers UNet / VAE names, OpenCLIP text-tower names (fused
 * in_proj rows are split into q/k/v), "conditioner.embedders.N.", "te1." … component aliases, llama.cpp-style T5 GGUF names. */
SD_API int64_t sd_load_weights_prefixed(sdm_ctx_t* ctx, const char* path, const char* prefix, int64_t* n_missing, int64_t* n_unused);
SD_API bool sd_convert_tensor_name(sdm_ctx_t* ctx, const char* name, char* out, size_t out_capacity); /* convert_tensor_name, name_conversion.cpp:1346 */

/* ---- the hot path ---- */
/* one diffusion-model forward (DiffusionModelRunner::compute, unet.hpp:818-858): x [W,H,C,N] f32,
 * timesteps [N], context [ctx_dim,n_tokens,N or 1], y [adm,N or 1] or NULL -> out [W,H,C,N] */
SD_API bool sd_unet_forward(sdm_ctx_t* ctx, const float* x, int w, int h, int c, int n, const float* timesteps,
                            const float* context, int64_t ctx_dim, int64_t n_tokens, int64_t ctx_n,
                            const float* y, int64_t y_dim, int64_t y_n, float* out);
/* the same forward WITHOUT the listed joint blocks — MMDiT::forward's skip_layers (mmdit.hpp:854-866), the extra evaluation of skip-layer guidance; MMDiT family only */
SD_API bool sd_unet_forward_skip_layers(sdm_ctx_t* ctx, const float* x, int w, int h, int c, int n, const float* timesteps, const float* context, int64_t ctx_dim,
                                        int64_t n_tokens, int64_t ctx_n, const float* y, int64_t y_dim, int64_t y_n, const int* skip_layers, int n_skip, float* out);
/* VAE decode_first_stage (stable-diffusion.cpp:3062-3078): latents [w,h,zc,n] (diffusion scale) -> rgb f32 [8w,8h,3,n] in [0,1] */
SD_API bool sd_vae_decode(sdm_ctx_t* ctx, const float* latents, int w, int h, int c, int n, float* out_rgb);
/* VAE encode — encode_first_stage (stable-diffusion.cpp:3042-3060): rgb f32 planar [w,h,3,n] in [0,1] -> diffusion latents [w/8,h/8,zc,n], SAMPLED from the encoder's diagonal
 * Gaussian with Philox(seed) like the reference (auto_encoder_kl.hpp:750-759) and scaled to the diffusion model's range; moments_out (optional) receives the graph's output
 * [w/8,h/8,2*zc,n] (mean | log-variance).  The encoder ("first_stage_model.encoder. ...", "first_stage_model.quant_conv. ...") is made on first use. */
SD_API bool sd_vae_encode(sdm_ctx_t* ctx, const float* rgb, int w, int h, int n, uint64_t seed, float* out_latents, float* moments_out);
/* sd_ctx_params_t::prediction: SDM_V_PRED switches the UNet families to CompVisVDenoiser's scalings (denoiser.hpp:1198-1205) in the host loop and the device-resident sampler */
SD_API bool sd_set_prediction(sdm_ctx_t* ctx, int prediction);
/* TAESD, the tiny autoencoder's decoder (src/model/vae/tae.hpp:123-183, 732-792; the reference's `--taesd`): the same latents -> rgb f32 [8w,8h,3,n], NOT clamped (the graph's
 * output is the image; the u8 stage clamps).  The module (parameters "tae.decoder.layers.<i>. ...", 4 latent channels, 16 for the DiT families) is made on first use;
 * sd_load_weights_prefixed(ctx, file, "tae.") loads a taesd checkpoint into it.  sd_use_tae(ctx, true): sdm_generate_image decodes with it instead of the KL-VAE. */
SD_API bool sd_tae_decode(sdm_ctx_t* ctx, const float* latents, int w, int h, int c, int n, float* out_rgb);
SD_API bool sd_use_tae(sdm_ctx_t* ctx, bool on);
/* sample(): init noise (Philox seed+b) -> Euler(-A) loop with CFG -> final latents [w,h,c,batch_count] */
SD_API bool sd_sample_latents(sdm_ctx_t* ctx, const sdm_img_gen_params_t* p, float* out_latents);
/* sdm_generate_image: sample + decode + uint8 RGB.  Caller frees with sdm_free_images (library callocs). */
SD_API bool sdm_generate_image(sdm_ctx_t* ctx, const sdm_img_gen_params_t* p, sdm_image_t** images_out, int* num_images_out);
SD_API void sdm_free_images(sdm_image_t* images, int num_images);

/* ---- host-side sampler pieces exposed for known-answer tests ---- */
SD_API void sd_philox_randn(uint64_t seed, uint32_t offset, uint32_t n, float* out); /* rng_philox.hpp:101-122 */
SD_API void sd_philox_uint32(uint64_t seed, uint32_t offset, uint32_t n, uint32_t* out /* 4*n words */); /* the integer stage alone: philox4_32, rng_philox.hpp:63-77 */
SD_API int sd_get_sigmas(int steps, float* out /* steps+1 */);                       /* denoiser.hpp:32-54 + stable-diffusion.cpp:173-186 */
/* CFG-pair split across two GPUs (the cond / uncond evaluations of src/runtime/guidance.cpp:149-179 on two devices):
 * with an exchange installed, the device-resident trajectory (device_sampler) evaluates ONE branch per step on this context — branch 0 = cond,
 * 1 = uncond — writes weight * eps (weight = cfg on branch 0, 1 - cfg on branch 1) to a device buffer and calls `exchange` once per step with
 * that buffer's DEVICE address, its f32 element count and the hipStream_t the producer graph was enqueued on (NULL on host backends).  The
 * callee sums the two ranks' buffers in place (one all-reduce, e.g. RCCL over one xGMI link) ordered on that stream, and returns false on
 * failure.  Nothing is copied to the host.  fn = NULL restores the single-device CFG pair. */
typedef bool (*sd_pair_exchange_fn)(void* device_eps, int64_t count, void* stream, void* user);
SD_API void sd_set_pair_exchange(sdm_ctx_t* ctx, sd_pair_exchange_fn fn, void* user, int branch);
/* Native RCCL form of the exchange (csrc/host/rccl_exchange.cpp): librccl.so is loaded with dlopen, no torch involved.  Rank 0 of a pair makes the
 * 128-byte unique id and ships it to its partner by any means (a file, a socket, torch.distributed in the tests); both create a 2-rank
 * communicator bound to HIP device `device` and install it: the engine then issues ONE in-place ncclAllReduce(SUM, f32) of its eps buffer per
 * sampler step on the backend stream (replaces the reference's host-side guidance.cpp:149-179 when cond and uncond run on two GPUs). */
SD_API bool sd_rccl_get_unique_id(void* id128);
SD_API void* sd_rccl_comm_create(int device, int nranks, int rank, const void* id128); /* NULL on failure (sd_rccl_last_error) */
SD_API void sd_rccl_comm_destroy(void* comm);
SD_API bool sd_set_pair_exchange_rccl(sdm_ctx_t* ctx, void* comm, int branch); /* comm = NULL removes the exchange */
SD_API const char* sd_rccl_last_error(void);
SD_API void sd_set_guidance(sdm_ctx_t* ctx, float guidance); /* FLUX distilled-guidance input (default 3.5, stable-diffusion.h guidance.distilled_guidance) */
/* the host sampler's classifier-free-guidance combine on n floats: out = uncond + scale * (cond - uncond), three separately rounded f32 operations like
 * sd::guidance::ClassifierFreeGuidance::forward on sd::Tensor<float> (src/runtime/guidance.cpp:171) — bit-exact against that code compiled into oracle/_ref */
SD_API void sd_cfg_combine(const float* cond, const float* uncond, int64_t n, float scale, float* out);
/* the pixel stage of sdm_generate_image on caller memory: planar CHW floats in [0, 1] -> interleaved RGB bytes (src/runtime/preprocessing.hpp:27-60); for the
 * bit-exact test against that header compiled into oracle/_ref */
SD_API void sd_planar_rgb_to_u8(const float* chw, int width, int height, uint8_t* out);
/* the host sampler loop (sigma ladder, initial noise, scalings / timestep per step, ancestral step, update, Philox noise order) on one image of n floats with
 * a synthetic model, denoised = x / (1 + sigma) + 0.01 * sigma: family 0 CompVis (SD1.x / SDXL), 1 discrete flow (SD3.x), 2 FLUX flow; method = sdm_sample_method_t;
 * eta INFINITY = the method's default; aux: optional 5 floats per step (c_skip, c_out, c_in, t, sigma).  Returns the ladder length, -1 on bad arguments.  For the
 * bit-exact tests against the reference's src/runtime/denoiser.hpp compiled into oracle/_ref (tests/test_host_logic.py). */
SD_API int sd_sample_synthetic(int family, int steps, int image_seq_len, int64_t n, uint64_t seed, int method, float eta, float* out, float* aux);
/* the same with a scheduler (sdm_scheduler_t; SDM_SCHEDULER_COUNT = the default) and any implemented method; aux receives 5 floats per MODEL CALL in call order (at most aux_calls
 * of them); returns the number of model calls, or -1 */
SD_API int sd_sample_synthetic2(int family, int steps, int image_seq_len, int64_t n, uint64_t seed, int method, int scheduler, float eta, float* out, float* aux, int aux_calls);
/* sigma ladder of a denoiser family (0 CompVis SD1.x, 1 discrete flow, 2 FLUX flow, 3 CompVis with the SDXL Align-Your-Steps table) under a scheduler; shift <= 0: the family's default;
 * returns the count (steps + 1) or -1 for a scheduler that is not implemented */
SD_API int sd_get_sigmas_sched(int family, int scheduler, int steps, int image_seq_len, float shift, float* out);
/* adaptive projected guidance over `steps` successive calls on one image of n floats (cond / uncond / out: [steps][n]); for the bit-exact test against guidance.cpp */
SD_API void sd_apg_sequence(const float* cond, const float* uncond, int64_t n, int steps, float scale, float eta, float momentum, float norm_threshold,
                            float norm_threshold_smoothing, float* out);
/* AutoEncoderKL::set_conv2d_scale (src/model/vae/auto_encoder_kl.hpp:708-717): every Conv2d of the VAE computes conv(x * s) / s + bias.  SDXL contexts start with
 * s = 1/32 — what the reference sets when no external VAE is given (src/stable-diffusion.cpp:1477-1485; `--vae` with a fixed VAE -> call this with 1) */
SD_API bool sd_set_vae_conv2d_scale(sdm_ctx_t* ctx, float scale); /* false (sd_last_error) for a non-finite or non-positive scale; loading a file under the
                                                                     "first_stage_model" prefix (= --vae) resets an SDXL context to scale 1 like the reference */
SD_API int sd_get_flux_sigmas(int steps, int image_seq_len, float* out /* steps+1 */); /* FluxScheduler, denoiser.hpp:726-782 */
SD_API int sd_gen_flux_pe(int h, int w, int patch_size, int context_len, const int* axes_dim, int n_axes, float theta, float* out); /* Rope::gen_flux_pe; returns floats written */
SD_API int sd_get_flow_sigmas(int steps, float shift, float* out /* steps+1 */);    /* DiscreteFlowDenoiser, denoiser.hpp:1239-1283 (t = 1000*sigma) */
SD_API float sd_sigma_to_t(float sigma);                                             /* denoiser.hpp:1140-1165 */

/* Node-by-node evaluation hook: the reference's sd_graph_eval_callback_t / sd_set_backend_eval_callback (include/stable-diffusion.h:442-447; used by the
 * imatrix collector, src/runtime/imatrix.cpp:39-100,180).  With a callback installed every graph a context computes goes through
 * sdm_backend_graph_compute_with_eval_callback, which restates sd_backend_graph_compute_with_eval_callback (src/core/ggml_extend_backend.cpp:466-509)
 * statement for statement: the callback is asked about every node (ask = true); the graph is cut BEHIND each node it wants, the slice is handed to the
 * backend as a SUB-GRAPH VIEW built exactly like sd_ggml_graph_view (:449-463: nodes + i0, n_leafs 0, leafs NULL, size 0, uid 0, the PARENT's use_counts /
 * visited_hash_set), computed async + synchronised, then the callback sees the node again (ask = false) and may read it and its sources; returning false
 * stops the graph (GGML_STATUS_ABORTED).  The sdm_ prefix keeps the names apart from the reference's own symbols. */
struct ggml_tensor;
struct ggml_cgraph;
struct ggml_backend;
typedef bool (*sdm_graph_eval_callback_t)(struct ggml_tensor* t, bool ask, void* user_data);
SD_API void sdm_set_backend_eval_callback(sdm_graph_eval_callback_t cb, void* user_data);
SD_API int sdm_backend_graph_compute_with_eval_callback(struct ggml_backend* backend, struct ggml_cgraph* gf, sdm_graph_eval_callback_t cb, void* user_data); /* enum ggml_status */

/* Graph topology as text, for the node-for-node comparison with the graphs the REFERENCE's own builders emit (tests/test_ref_graphs.py; the reference side is
 * oracle/ref_graphs_wrap.cpp -> oracle/_ref/libref_graphs.so, which calls this same function on its graphs).  One line per leaf ("L <j> type ne nb flags name")
 * and per node ("N <i> OP type ne nb op_params flags sources view_src@offset name"), sources as n<i> / l<j>.  Returns the bytes needed incl. the
 * terminating 0 (call with cap 0 to size the buffer).  sdm_set_graph_capture(1): every graph an engine context computes is described just before it is
 * submitted and kept until the next one; 2: described and NOT computed (the call fails with "graph captured, compute skipped": topology of full-size models
 * without the arithmetic); sdm_last_graph_description copies it out. */
SD_API size_t sdm_graph_describe(struct ggml_cgraph* gf, char* buf, size_t cap);
SD_API void sdm_set_graph_capture(int on);
SD_API size_t sdm_last_graph_description(char* buf, size_t cap);

/* ---- timing / introspection ---- */
typedef struct {
    double last_sample_ms;  /* denoise loop wall time of the last sd_sample_latents / sdm_generate_image */
    double last_decode_ms;  /* VAE decode wall time */
    int64_t unet_calls;     /* graph computes issued */
    int64_t graph_nodes;    /* nodes in the last UNet graph */
    size_t compute_buffer_bytes;
    size_t weight_bytes;
    /* cumulative HOST time of the denoiser runner since context creation (ms): graph construction, gallocr allocation, and
     * uploads + graph_compute (which contains the device time on the synchronous path, only the enqueue cost on the device-resident path) */
    double host_build_ms, host_alloc_ms, host_submit_ms;
    int64_t graph_cache_hits; /* denoiser calls that replayed the cached graph (same shapes as the previous call) instead of rebuilding it */
} sd_stats_t;
SD_API void sd_get_stats(sdm_ctx_t* ctx, sd_stats_t* out);
/* ---- text encoders + conditioner --------------------------------------------------------
 * CLIP text towers (src/model/te/clip.hpp) and the T5 encoder (src/model/te/t5.hpp) as graphs on the same backend, and the
 * conditioner composition of src/conditioning/conditioner.hpp (SD1.x/SDXL :414-544, SD3 :842-1015, FLUX :1209-1297).  Inputs are
 * TOKEN IDS (+ per-token prompt weights): the reference's vocabularies are not part of its source drop, so tokenisation stays with
 * the caller.  Encoders are created on first use with synthetic weights (weight_seed); sd_load_weights overwrites them when the file
 * names "cond_stage_model.*" / "text_encoders.*" tensors. */
typedef struct {
    const int32_t* ids;
    const float* weights; /* NULL = all 1.0 */
    int n;                /* multiple of the chunk length: 77 for CLIP and SD3's T5, 256 for FLUX's T5 */
} sd_token_list_t;
SD_API bool sd_text_encoders_init(sdm_ctx_t* ctx);
/* which: 0 = clip_l (ViT-L/14), 1 = clip_g (ViT-bigG/14).  CLIPTextModelRunner::compute (clip.hpp:562-583): hidden states
 * [hidden, n_tokens] after layer n_layer - clip_skip (clip_skip <= 0: all layers), or with return_pooled the final-LN'd row
 * max_token_idx (times text_projection for bigG).  Returns floats written, -1 on error. */
SD_API int64_t sd_clip_forward(sdm_ctx_t* ctx, int which, const int32_t* ids, int n_tokens, int max_token_idx, bool return_pooled, int clip_skip, float* out,
                               int64_t out_capacity);
SD_API int64_t sd_t5_forward(sdm_ctx_t* ctx, const int32_t* ids, int n_tokens, float* out, int64_t out_capacity); /* T5Runner::compute, t5.hpp:452-461 */
SD_API int sd_t5_relative_position_buckets(int q_len, int k_len, int32_t* out /* q_len*k_len */);                 /* t5.hpp:463-530 */
/* token ids -> SDCondition.  crossattn_ne = {ctx_dim, n_tokens}; pass NULL output pointers to query the sizes first.  clip_g / t5
 * are ignored by families that do not own them (SD1.x and SDXL derive the bigG ids from clip_l like the reference). */
SD_API bool sd_get_learned_condition(sdm_ctx_t* ctx, const sd_token_list_t* clip_l, const sd_token_list_t* clip_g, const sd_token_list_t* t5, int clip_skip,
                                     int width, int height, bool zero_out_masked, float* crossattn_out, int64_t crossattn_capacity, int64_t* crossattn_ne,
                                     float* vector_out, int64_t vector_capacity, int64_t* vector_n);


#ifdef __cplusplus
}
#endif
#endif
