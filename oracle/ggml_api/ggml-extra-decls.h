/* ggml-extra-decls.h — TEST INFRASTRUCTURE (oracle/Makefile target _ref/libref_graphs.so).
 *
 * The reference's graph builders (src/core/ggml_extend.hpp, src/model/common/block.hpp, src/model/diffusion/{unet,mmdit,flux,dit}.hpp,
 * src/model/vae/auto_encoder_kl.hpp) are compiled from where they lie against THIS REPOSITORY'S ggml front-end (stable-diffusion.cpp_amd/csrc/ggml:
 * the clean-room implementation of ggml's public graph-construction API the product host links).  Those headers also MENTION ggml API the hot path never
 * calls — the backend scheduler, 3-D / depthwise convolutions, the fork's int8 ops, logging — inside functions that are parsed but not executed when a
 * UNet / MMDiT / FLUX / VAE graph is built.  They are DECLARED here (upstream signatures as far as the call sites fix them); oracle/ref_graphs_wrap.cpp
 * defines each as a function that aborts with its own name, so reaching one of them cannot silently compute something.  Nothing here stands in for ggml's arithmetic: libref_graphs.so only BUILDS graphs (node sequences, shapes, op_params, names),
 * which the tests compare node for node with the graphs csrc/host/models.hpp emits and then run through the backends. */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum ggml_log_level {
    GGML_LOG_LEVEL_NONE  = 0,
    GGML_LOG_LEVEL_DEBUG = 1,
    GGML_LOG_LEVEL_INFO  = 2,
    GGML_LOG_LEVEL_WARN  = 3,
    GGML_LOG_LEVEL_ERROR = 4,
    GGML_LOG_LEVEL_CONT  = 5,
};
typedef void (*ggml_log_callback)(enum ggml_log_level level, const char* text, void* user_data);
void ggml_log_set(ggml_log_callback log_callback, void* user_data);

/* graph-construction API the hot path does not reach */
struct ggml_tensor* ggml_pad_ext_circular(struct ggml_context* ctx, struct ggml_tensor* a, int lp0, int rp0, int lp1, int rp1, int lp2, int rp2, int lp3, int rp3);
struct ggml_tensor* ggml_roll(struct ggml_context* ctx, struct ggml_tensor* a, int shift0, int shift1, int shift2, int shift3);
struct ggml_tensor* ggml_interpolate(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, uint32_t mode);
struct ggml_tensor* ggml_arange(struct ggml_context* ctx, float start, float stop, float step);
struct ggml_tensor* ggml_im2col_3d(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int64_t IC, int s0, int s1, int s2, int p0, int p1, int p2,
                                   int d0, int d1, int d2, enum ggml_type dst_type);
struct ggml_tensor* ggml_conv_3d(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int64_t IC, int s0, int s1, int s2, int p0, int p1, int p2,
                                 int d0, int d1, int d2);
struct ggml_tensor* ggml_conv_3d_direct(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int s0, int s1, int s2, int p0, int p1, int p2, int d0,
                                        int d1, int d2, int n_channels, int n_batch, int n_channels_out);
struct ggml_tensor* ggml_conv_2d_dw(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1);
struct ggml_tensor* ggml_conv_2d_dw_direct(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int stride0, int stride1, int pad0, int pad1,
                                           int dilation0, int dilation1);
/* ops of the reference's ggml fork (src/core/ggml_extend.hpp:1059, 3492) */
struct ggml_tensor* ggml_mul_mat_i8_tensorwise(struct ggml_context* ctx, struct ggml_tensor* w, struct ggml_tensor* x, struct ggml_tensor* weight_scale,
                                               struct ggml_tensor* bias, int convrot_group_size);
struct ggml_tensor* ggml_l2_norm(struct ggml_context* ctx, struct ggml_tensor* a, float eps);
struct ggml_tensor* ggml_quantize_i8_convrot(struct ggml_context* ctx, struct ggml_tensor* x, int group_size);

/* the backend scheduler (GGMLRunner's multi-device path, src/core/ggml_extend.hpp:2090-2250, 2820-2865): never created by the graph-topology wrapper */
typedef struct ggml_backend_sched* ggml_backend_sched_t;
ggml_backend_sched_t ggml_backend_sched_new(ggml_backend_t* backends, ggml_backend_buffer_type_t* bufts, int n_backends, size_t graph_size, bool parallel, bool op_offload);
void ggml_backend_sched_free(ggml_backend_sched_t sched);
void ggml_backend_sched_reset(ggml_backend_sched_t sched);
void ggml_backend_sched_synchronize(ggml_backend_sched_t sched);
bool ggml_backend_sched_alloc_graph(ggml_backend_sched_t sched, struct ggml_cgraph* graph);
enum ggml_status ggml_backend_sched_graph_compute(ggml_backend_sched_t sched, struct ggml_cgraph* graph);
void ggml_backend_sched_set_tensor_backend(ggml_backend_sched_t sched, struct ggml_tensor* node, ggml_backend_t backend);

/* registry / device-selection helpers and proc-address typedefs src/core/ggml_extend_backend.cpp mentions (backend selection by name, CPU thread count,
 * feature listing, row-split buffers): the wrapper is handed a backend that already exists */
struct ggml_backend_feature {
    const char* name;
    const char* value;
};
typedef void (*ggml_backend_set_n_threads_t)(ggml_backend_t backend, int n_threads);
typedef struct ggml_backend_feature* (*ggml_backend_get_features_t)(ggml_backend_reg_t reg);
typedef ggml_backend_buffer_type_t (*ggml_backend_split_buffer_type_t)(int main_device, const float* tensor_split);
/* graph-cut segmented execution (src/core/ggml_graph_cut.cpp:640-740): only entered when a graph does not fit the device (never at 288 GB) */
void ggml_gallocr_reserve_n_size(ggml_gallocr_t galloc, struct ggml_cgraph* graph, const int* node_buffer_ids, const int* leaf_buffer_ids, size_t* sizes);

#ifdef __cplusplus
}
#endif
