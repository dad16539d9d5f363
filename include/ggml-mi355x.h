/*
 * ggml-mi355x.h — C ABI exported by libggml-mi355x.so, the MI355X (gfx950) ggml backend plug-in.
 *
 * The library is a drop-in for the reference's ggml-backend compute path: a host that was built with
 * GGML_BACKEND_DL (the reference's CI does exactly this: .github/workflows/build.yml:88,686) discovers
 * it with ggml_backend_load_all(), enumerates its devices and drives it ONLY through the vtables of
 * include/ggml-abi.h.  Each entry point below cites the reference interface it plugs into.
 *
 *   ggml_backend_init()            <- dlsym'd by ggml_backend_load()/load_all()
 *                                     (src/core/ggml_extend_backend.cpp:302-320 -> ggml_backend_load_all)
 *   ggml_backend_score()           <- optional ranking among several libggml-*.so variants
 *   ggml_backend_mi355x_reg()      <- static-registration form (what ggml-backend-reg.cpp would call if
 *                                     the backend were compiled into ggml: `register_backend(ggml_backend_mi355x_reg())`)
 *
 * Reached through the returned registry (struct ggml_backend_reg, include/ggml-abi.h):
 *   reg.get_device(i)              <- ggml_backend_dev_get / dev_by_name (ggml_extend_backend.cpp:334-418)
 *   dev.supports_op(node)          <- probed at graph BUILD time (ggml_extend.hpp:1425, :2198-2210)
 *   dev.get_buffer_type / buft.alloc_buffer / buffer.set_tensor,get_tensor
 *                                  <- gallocr + weight staging (ggml_extend.hpp:2227-2245, 2347-2435;
 *                                     model_manager.cpp:470-477, 735-750)
 *   dev.init_backend -> backend.graph_compute(cgraph)
 *                                  <- THE hot-path call, src/core/ggml_extend_backend.cpp:471 (sync form)
 *                                     and :489 (async on a sub-graph VIEW: n_nodes slice, leafs == NULL)
 *   backend.synchronize            <- ggml_extend_backend.cpp:494,504; ggml_extend.hpp:1526,2330
 *   reg.get_proc_address(name)     <- "ggml_backend_split_buffer_type" / "ggml_backend_set_n_threads" /
 *                                     "ggml_backend_get_features" -> NULL (host handles absence,
 *                                     ggml_extend_backend.cpp:801-804,440-446,519-528);
 *                                     "ggml_backend_mi355x_get_stats" / "_set_option" -> the functions below
 *
 * Error behaviour (SURVEY.md §8(b) Errors): graph_compute returns GGML_STATUS_FAILED for an unsupported
 * node or a HIP launch error and GGML_STATUS_ALLOC_FAILED when an internal workspace cannot be
 * allocated — it never aborts; alloc_buffer returns NULL on hipMalloc failure.
 */
#ifndef GGML_MI355X_H
#define GGML_MI355X_H

#include "ggml-abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MI355X_API __attribute__((visibility("default")))

GGML_MI355X_API ggml_backend_reg_t ggml_backend_init(void);
GGML_MI355X_API int ggml_backend_score(void);
GGML_MI355X_API ggml_backend_reg_t ggml_backend_mi355x_reg(void);
GGML_MI355X_API int ggml_backend_mi355x_get_device_count(void);

/* execution statistics of the graph planner (process-wide, monotonically increasing) */
struct ggml_backend_mi355x_stats {
    int64_t graphs_computed;     /* graph_compute calls */
    int64_t plans_built;         /* plan-cache misses (topology seen for the first time) */
    int64_t nodes_seen;          /* ggml nodes in the graphs of plans built */
    int64_t kernels_planned;     /* kernel launches in plans built (after fusion) */
    int64_t kernels_launched;    /* kernel launches issued */
    int64_t fused_conv;          /* IM2COL+MUL_MAT+CONT(+ADD...) chains replaced by the implicit-GEMM conv */
    int64_t fused_conv_bounced;  /* ... of which had to bounce through scratch because dst aliased the input */
    int64_t fused_linear;        /* MUL_MAT(+bias)(+residual) on the MFMA weight GEMM */
    int64_t fused_norm;          /* GROUP_NORM/NORM + affine (+SiLU) */
    int64_t fused_geglu;
    int64_t fused_attention;     /* attention sub-graphs routed to the flash kernel */
    int64_t generic_matmul;      /* MUL_MAT on the generic exact-f32 path */
    int64_t swizzled_weight_bytes;
    int64_t graph_replays;       /* hipGraph replays (when graph capture is enabled) */
    int64_t fused_linear_geglu;  /* FF1 GEMM + GEGLU in one kernel (the [tokens][2*inner] tensor is never written) */
    int64_t split_k_gemms;       /* gemm16 contractions planned with a split-K workspace */
    int64_t head_major_gemms;    /* q/k/v projections that store the attention operand layout directly */
    int64_t fused_modulate;      /* LayerNorm + adaLN modulate written straight to the next GEMM's f16 operand image (DiT) */
    int64_t fused_gate;          /* Linear -> * gate -> + x folded into the GEMM epilogue (DiT) */
    int64_t fused_gelu;          /* fc1 -> GELU written as fc2's f16 operand image */
    int64_t fused_rope;          /* Rope::apply_rope node chains (cont/repeat/mul/add) replaced by one rotary kernel */
    int64_t fused_concat_heads;  /* token concat + head-major permute (+ f16 cast) of the MMDiT joint-attention operands in one pass */
    int64_t qgemv_linears;       /* q8_0 / q4_0 Linears planned on the in-register dequant kernel (raw blocks streamed, no f16 weight image) */
    int64_t fused_chan_add;      /* conv + ADD(time-embedding [1,1,C,N]) folded into the conv epilogue (ResBlock) */
    int64_t fused_proj_tokens;   /* SpatialTransformer 1x1 proj_in / proj_out run as token GEMMs (the NCHW <-> token transposes are not executed) */
    int64_t gemm_attention;      /* attention chains with head dims beyond the flash kernel (VAE d = 512) composed from MFMA GEMMs + f16 row softmax */
    int64_t fused_q16;           /* Q projections stored as the flash kernel's f16 head-major operand (the f32 CONT buffer is never written) */
    int64_t split_k_inlaunch;    /* split-K GEMMs whose slices are combined by the last-arriving workgroup of each output tile (no reduce pass) */
    int64_t qgemm16_linears;     /* q8_0 / q4_0 Linears with 3 .. qgemm16_max_rows activation rows on k_qgemm16: raw GGUF blocks streamed, dequantised in registers into MFMA fragments (no f16 weight image) */
    int64_t fgemv_linears;       /* f16 / f32 Linears with <= 16 activation rows on k_fgemv (one weight-streaming launch instead of SiLU + pack + GEMM + split-K reduce) */
    int64_t fused_presilu;       /* SiLU nodes in front of a few-row Linear applied by k_fgemv / k_qgemv while staging the rows (the UNARY node is not executed) */
    int64_t fused_sibling_linears; /* head-major projection GEMMs saved by running the q / k / v (or k / v) Linears of one attention as ONE launch over the shared operand */
    int64_t hoisted_kv_linears;  /* cross-attention K / V projections of the text context computed in grouped launches ahead of their graph position (results in the arena) */
    int64_t window_convs;        /* 3x3 convs planned on the LDS-window kernel (conv3w.hip) */
    int64_t hoisted_emb_linears; /* per-ResBlock SiLU(emb) -> Linear projections computed by one grouped weight-streaming launch ahead of their graph position */
    int64_t fused_rows16;        /* Linear (+bias, +residual) read only by a 1x1 conv (SpatialTransformer proj_out): written as the conv's f16 operand rows */
    int64_t fused_joint_qkv;     /* MMDiT streams whose fused qkv projection feeds the joint attention through k_joint_heads (no split copy, no separate norms / concats) */
    int64_t jit_images;          /* quantised Linears planned with a just-in-time f16 image (option jit_qimages: no cached image, rebuilt in front of every launch) */
    int64_t fused_cat_rows16;    /* CONCAT along the feature dimension read only by Linears: their f16 operand image assembled directly (FLUX single block) */
    int64_t fused_gn_stats;      /* split-K convs whose slab reduce also writes the statistics of the GroupNorm that reads the result (k_splitk_reduce_gn) */
    int64_t fused_ln_reduce;     /* split-K Linears whose slab reduce also writes the f16 operand image of the LayerNorm that reads the result (k_splitk_reduce_ln) */
    int64_t redirect_fallbacks;  /* graphs planned a second time without the joint-qkv pre-passes because a redirected projection was not taken by its Linear */
    int64_t fused_concat_gn;     /* skip-connection CONCATs never materialised: GroupNorm statistics / apply (and the skip conv's operand cast) read the two sources (plan_concat_gn) */
    int64_t fused_conv_scale;    /* Conv2d scales (SCALE s -> conv -> SCALE 1/s, the reference's SDXL VAE setting) folded into the operand image and the epilogue */
    int64_t view_graphs;         /* plans built for SUB-GRAPH VIEWS (sd_ggml_graph_view, src/core/ggml_extend_backend.cpp:449-463: leafs NULL / size 0) */
    int64_t plans_evicted;       /* cached plans (with their captured hipGraph) dropped by the LRU bound of the plan cache (option plan_cache_cap, default 512) */
    int64_t hoisted_mod_linears; /* DiT modulation Linears (FLUX Modulation / SD3 adaLN: one or two rows, raw q8_0 / q4_0 weights, same input vector) computed by ONE grouped weight-streaming launch ahead of their graph position */
    int64_t jit_overlapped;      /* just-in-time weight-image rebuilds issued one Linear ahead on the side stream, overlapping the previous GEMM (option jit_overlap) */
    int64_t view_external_nodes; /* nodes of those slices treated as read outside the slice (parent use_counts > readers inside, the slice's last node and its sources) */
    int64_t qinloop_linears;     /* q8_0 / q4_0 Linears above k_qgemm16's row range planned on the pipelined 256 x 256 tile with the raw GGUF blocks dequantised INSIDE the main loop (k_gemm16<..., QT>): no f16 weight image, resident or rebuilt */
    int64_t flash_out_alias;     /* FLASH_ATTN_EXT -> VIEW -> CONT chains NOT written by the flash kernel itself because the graph allocator gave the CONT the block of a Q / K / V operand (the node runs plain, the CONT as a copy) */
    int64_t flash_slice_images;  /* token-slice views of an attention output registered on the flash kernel's f16 operand image (their Linears read runs of rows out of it: no f32 tensor, no pack pass) */
};
GGML_MI355X_API void ggml_backend_mi355x_get_stats(struct ggml_backend_mi355x_stats* out);
/* Host enum numbering, resolved BY NAME.  The numeric values of `enum ggml_op` / `enum ggml_unary_op` in ggml-abi.h are a recollection of upstream, and
 * the reference links a fork that adds ops (src/core/ggml_extend.hpp:1059,1088,3492): ggml_backend_init() looks up the host's ggml_op_name /
 * ggml_unary_op_name / ggml_type_name (dlsym over the loaded objects) and rebuilds the planner's op tables from the names; type numbers (part of the
 * GGUF file format) are verified.  A host that lacks a needed name, names two numbers alike, or numbers a type differently gets ZERO devices and the
 * reason on stderr and in ggml_backend_mi355x_enum_status().  A host that hides those functions can call ggml_backend_mi355x_resolve_enums() itself
 * BEFORE ggml_backend_init(); returns 0, or -1 with the tables unchanged.  get_enum_maps: host number -> ggml-abi.h number (GGML_OP_COUNT /
 * GGML_UNARY_OP_COUNT = unknown to this backend), 256 entries each. */
GGML_MI355X_API int ggml_backend_mi355x_resolve_enums(const char* (*op_name)(int), const char* (*unary_op_name)(int), const char* (*type_name)(int));
GGML_MI355X_API const char* ggml_backend_mi355x_enum_status(void);
GGML_MI355X_API void ggml_backend_mi355x_get_enum_maps(uint8_t* ops256, uint8_t* unary256);
/* live per-kernel-family timing (bench.py's roofline legs): while a family's bit is enabled, every dispatch of that family is bracketed by
 * HIP events recorded on the launch stream.  get_* synchronise the device, return the totals since enable / the last get_, and reset.
 * Family indices (bit positions of `family_mask`): 0 conv 256-row tiles, 1 conv 128-row tiles, 2 Linear GEMM, 3 flash attention,
 * 4 q8_0/q4_0 in-register dequant GEMM, 5 f32 MFMA matmul, 6 k_nchw_to_nhwc_f16, 7 k_layer_norm_f16, 8 k_gn_stats, 9 f16 pack,
 * 10 copies / transposes, 11 binary elementwise, 12 concat, 13 unary / scale, 14 split-K reduce, 15 f32 norms, 16 softmax, 17 other. */
struct ggml_backend_mi355x_kernel_timing {
    char kernel[96];     /* kernel family the events bracket */
    int64_t launches;
    double total_ms;     /* sum of hipEventElapsedTime over the launches */
    double total_flops;  /* sum of the launches' algorithmic FLOPs (conv: 2 * positions * IC*KH*KW * OC; attention: 4 * Lq * Lk * heads * d) */
    double total_bytes;  /* sum of the launches' algorithmic HBM bytes (bandwidth-bound families: one read + one write of the activation) */
    int32_t bound;       /* 0 = judged against the MFMA peak, 1 = against HBM bandwidth */
    int32_t family;
};
GGML_MI355X_API void ggml_backend_mi355x_kernel_timing_enable(int enable);           /* != 0: family 0 only (the dominant kernel; cheap enough for the timed region) */
GGML_MI355X_API void ggml_backend_mi355x_kernel_timing_enable_mask(uint32_t family_mask);
GGML_MI355X_API void ggml_backend_mi355x_get_kernel_timing(struct ggml_backend_mi355x_kernel_timing* out);  /* the first timed family */
GGML_MI355X_API int ggml_backend_mi355x_get_kernel_timings(struct ggml_backend_mi355x_kernel_timing* out, int capacity);  /* every family with launches; returns the count */
/* options (default): "fusion" (1), "mfma_gemm" (1), "hip_graph" (1: a plan is captured into a hipGraph the second time it runs and replayed from then on; 2 = capture at the first run; eager while kernel timing is on), "flash_pattern" (1), "gemm16" (1), "gemm16_variant" (3), "gemm16_tile" (-1),
 * "splitk_target" (384), "conv_tap_major" (0), "fuse_modulate" / "fuse_gate" / "fuse_gelu" / "fuse_rope" / "fuse_concat_heads" (1);
 * "splitk_mid" (0: two K slices for launches of 193..384 workgroups), "pinned_uploads" (0: set_tensor_async stages through pinned host memory so the call does
 * not wait for the stream);
 * few-row / quantised Linears: "fgemv" (1: f16 / f32 weights under <= 16 rows on the one-launch weight-streaming kernel, SiLU in front of it applied on
 * load), "fgemv_max_rows" (16), "qgemv" (1) / "qgemv_max_rows" (4, <= 16: raw q8_0 / q4_0 blocks streamed up to that many rows), "qgemm16_max_rows" (512:
 * raw-block MFMA GEMM up to n rows; 8192 = the resident-quantised mode, no f16 image for any quantised Linear), "qgemm16" (1);
 * "fuse_linear_nchw" (1: Linear proj_out -> PERMUTE -> CONT -> RESHAPE -> ADD x_in of such a SpatialTransformer runs as a 1x1 implicit-GEMM conv over the token rows: NCHW + bias + residual epilogue),
 * "fuse_gn_tokens" (1: GroupNorm -> PERMUTE -> CONT -> Linear proj_in of a SpatialTransformer with Linear projections (SDXL): the GroupNorm apply pass writes the Linear's f16 operand image),
 * "fuse_concat_gn" (1: UNet skip-connection CONCAT read only by a GroupNorm chain (+ the skip 1x1 conv) is never built, option 0 = the concat pass),
 * "fuse_split_gelu" (1: FLUX linear1 writes gelu(mlp) as f16 into linear2's operand image), "geglu16" (1: GEGLU FF1 on the 256 x 320 tile through the 16-column
 * value / gate interleave), "streamk" (0; 1 / 2 / 3: stream-K policies for the pipelined 256 x 256 Linear tile), "qgemm16_rb" (3), "conv3w_prio" (3);
 * "relax_res_overlap" (1: a Linear + residual ADD fuses even when the allocator put the sum on the Linear input's released f32 buffer — GEMM launches read the
 * arena's f16 operand image, not that buffer; 0 = the round-3 test);
 * launch grouping: "fuse_siblings" (1: q / k / v projections of one attention as one multi-weight launch), "hoist_kv" (1: cross-attention K / V
 * projections of all blocks grouped ahead of their graph position, results in the arena); "fuse_q16", "fuse_chan_add", "fuse_proj_tokens" (1);
 * "hoist_emb" (1: the per-ResBlock SiLU(emb) -> Linear projections as one grouped weight-streaming launch), "fuse_joint_qkv" (1: MMDiT joint attention — qkv projections into arena scratch, split / per-head RMSNorm / token
 * concat / head-major cast as one pass per operand), "fuse_cat_rows16" (1: concat(a, b) along features feeding only Linears is assembled as their f16
 * operand image — flash output and gelu(strided view) write their columns themselves), "fuse_gn_stats" (1: the slab reduce of a split-K conv
 * also computes the statistics of the GroupNorm that reads its result), "fuse_rows16" (0: a Linear read only by a
 * 1x1 conv writes the conv's f16 operand rows; measured slightly slower on SD1.5);
 * "splitk_inkernel" (0: split-K combined by the last-arriving workgroup, 128-row tiles; measured slower) / "splitk_in_target" (320);
 * conv: "conv3w" (1: 3x3 / stride-1 convs on 16..128-wide maps on the LDS-window kernel), "conv3w_min_blocks" (8) / "conv3w_min_blocks_deep" (5:
 * least 32-channel blocks per K slice when the window kernel splits K);
 * round 5: "fuse_conv_scale" (1: Conv2d scale — SCALE s -> conv -> SCALE 1/s, the reference's SDXL VAE setting — folded into the operand image and the epilogue),
 * "conv_wmajor" (1: weight-major workgroup order for convs whose weight image is >= 2x their input image: every (column tile, K slice) weight chunk on one XCD),
 * "t256p_pad" (1: the pipelined 256 x 256 Linear tile also for widths that are multiples of 128 only), "tail_split" (0: row-split launches — whole rounds of 256 x 256
 * tiles + the remaining rows on small tiles; measured neutral), "ln16_rows" (1; 4 = four rows per wave in the LayerNorm -> f16 image kernel: measured slower);
 * "gemm16_swp" (0; 1 = the 256-row Linear tiles with the accumulator transposed, 16-byte epilogue accesses: correct, measured 1 % slower per SD1.5 step);
 * round 6: "qinloop_min_rows" (513: q8_0 / q4_0 Linears with at least that many activation rows whose launch takes the pipelined 256 x 256 tile read the RAW GGUF blocks and
 * dequantise them inside the GEMM's main loop — no f16 weight image, cached or rebuilt; 0 = off: cached image, or "jit_qimages" rebuild), "qgemm16_pf" (1; 2 = two
 * segments of loads in flight in k_qgemm16: measured slower), "fuse_flash_slices" (1: the proj Linears of an MMDiT / FLUX double block read their token slices out of the flash kernel's f16 image), "gemm16_t192p" (1: the pipelined 256 x 192 Linear tile where it quantises better on 256 CUs), "hoist_mod" (1), "jit_overlap" (0), "plan_cache_cap" (512), "gn_split_min" (65536);
 * "fuse_ln_reduce" (1: the slab reduce of a split-K Linear also writes the f16 operand image of the LayerNorm that reads its result);
 * flash attention: "flash_vtr" (31: bit per head-dim class — V tiles row-major in LDS, fragments by ds_read_b64_tr_b16; 0 = transposing staging pass),
 * "flash_ovl" (1: the two-block d = 40 kernel issues one block's softmax inside the other block's MFMAs; 2: also the other d <= 48 launches; 0: off),
 * "flash_nsel" (1: select-free K / V staging, bit-identical; 0 = per-chunk selects), "flash_short" (2: register-resident K / V kernel for the
 * 77-token cross-attentions, 64 < Lk <= 96 and d <= 64, with the next block's Q rows prefetched; 1 = without the prefetch; 0 = the tile kernel),
 * "flash_qb2" (1: two query blocks per wave for d <= 48), "flash_pp" (0; 1 = the 8-wave ping-pong kernel for 64 < d <= 96,
 * 2 = wherever it is legal: measured slower or equal, kept for A/B runs), "flash_pp_min_tiles" (4).
 * Wrong-result timing ablations exist only in builds with -DMI355X_EXPERIMENTS ("flash_ablate"). */
GGML_MI355X_API void ggml_backend_mi355x_set_option(const char* key, int value);
/* What the calling thread's current device delivers, measured in about a second (csrc/kernels/calib.hip): an MFMA loop from registers (f16, 32x32x16: the
 * matrix pipe's ceiling at the clock the chip sustains under it, and that clock), a float4 copy and a read-only pass over 1 GiB.  bench.py reports them as
 * roofline.measured_peaks next to the vendor peaks, so that runs on boxes of different speed can be compared.  Returns 0, or -1 on failure. */
struct ggml_backend_mi355x_calibration {
    float mfma_f16_tflops; /* dense f16 MFMA, TFLOP/s */
    float mfma_clock_mhz;  /* shader clock during the MFMA loop */
    float copy_tbs;        /* (read + write bytes) / time of a float4 copy, TB/s */
    float read_tbs;        /* read-only pass, TB/s */
    int compute_units;
};
GGML_MI355X_API int ggml_backend_mi355x_calibrate(struct ggml_backend_mi355x_calibration* out);
/* the hipStream_t every graph of this backend instance is enqueued on (graph_compute_async, set/get_tensor_async): a caller that touches a
 * tensor's device memory between two graphs (the CFG-pair all-reduce, sd_set_pair_exchange) orders its work on this stream */
GGML_MI355X_API void* ggml_backend_mi355x_get_stream(ggml_backend_t backend);
/* path of the HIP runtime library this plug-in is bound to, and hipSetDevice through it (for companions that must share its streams: RCCL) */
GGML_MI355X_API const char* ggml_backend_mi355x_hip_library(void);
GGML_MI355X_API int ggml_backend_mi355x_set_device(int hip_device);
GGML_MI355X_API int ggml_backend_mi355x_get_device(void); /* the calling thread's current HIP device of the plug-in's runtime, -1 on error */

#ifdef __cplusplus
}
#endif
#endif
