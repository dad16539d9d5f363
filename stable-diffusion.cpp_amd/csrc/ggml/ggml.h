/*
 * ggml.h — mini graph front-end: the subset of ggml's PUBLIC API that stable-diffusion.cpp's hot path
 * uses to build, allocate and dispatch compute graphs (reference: every `ggml_*` call in
 * src/core/ggml_extend.hpp:953-1652,2023-2930 and src/model/{common,diffusion,vae} headers).
 *
 * Clean-room: the reference's ggml submodule is absent; function names, argument order,
 * result shapes and op_params encodings follow upstream ggml so that host code written against this
 * header compiles unchanged against the real library.  Only metadata lives here — ALL arithmetic runs
 * behind the backend interface declared in include/ggml-abi.h.
 */
#pragma once
#include "ggml-abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_API __attribute__((visibility("default")))

struct ggml_init_params {
    size_t mem_size;   /* ignored (objects are heap-allocated individually) */
    void* mem_buffer;  /* ignored */
    bool no_alloc;     /* true: tensors get no host data (backend buffers hold it) */
};

/* ---- context / tensors ---- */
GGML_API struct ggml_context* ggml_init(struct ggml_init_params params);
GGML_API void ggml_free(struct ggml_context* ctx);
GGML_API bool ggml_get_no_alloc(struct ggml_context* ctx);
GGML_API size_t ggml_tensor_overhead(void);
GGML_API size_t ggml_graph_overhead(void);
GGML_API size_t ggml_graph_overhead_custom(size_t size, bool grads);

GGML_API struct ggml_tensor* ggml_new_tensor(struct ggml_context* ctx, enum ggml_type type, int n_dims, const int64_t* ne);
GGML_API struct ggml_tensor* ggml_new_tensor_1d(struct ggml_context* ctx, enum ggml_type type, int64_t ne0);
GGML_API struct ggml_tensor* ggml_new_tensor_2d(struct ggml_context* ctx, enum ggml_type type, int64_t ne0, int64_t ne1);
GGML_API struct ggml_tensor* ggml_new_tensor_3d(struct ggml_context* ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2);
GGML_API struct ggml_tensor* ggml_new_tensor_4d(struct ggml_context* ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
GGML_API struct ggml_tensor* ggml_dup_tensor(struct ggml_context* ctx, const struct ggml_tensor* src);
GGML_API struct ggml_tensor* ggml_view_tensor(struct ggml_context* ctx, struct ggml_tensor* src);
GGML_API struct ggml_tensor* ggml_get_first_tensor(const struct ggml_context* ctx);
GGML_API struct ggml_tensor* ggml_get_next_tensor(const struct ggml_context* ctx, struct ggml_tensor* tensor);
GGML_API struct ggml_tensor* ggml_get_tensor(struct ggml_context* ctx, const char* name);

GGML_API struct ggml_tensor* ggml_set_name(struct ggml_tensor* tensor, const char* name);
GGML_API const char* ggml_get_name(const struct ggml_tensor* tensor);
GGML_API void ggml_set_input(struct ggml_tensor* tensor);
GGML_API void ggml_set_output(struct ggml_tensor* tensor);
GGML_API void ggml_set_param(struct ggml_tensor* tensor);

GGML_API int64_t ggml_nelements(const struct ggml_tensor* tensor);
GGML_API int64_t ggml_nrows(const struct ggml_tensor* tensor);
GGML_API size_t ggml_nbytes(const struct ggml_tensor* tensor);
GGML_API int64_t ggml_blck_size(enum ggml_type type);
GGML_API size_t ggml_type_size(enum ggml_type type);
GGML_API size_t ggml_row_size(enum ggml_type type, int64_t ne);
GGML_API size_t ggml_element_size(const struct ggml_tensor* tensor);
GGML_API const char* ggml_type_name(enum ggml_type type);
GGML_API const char* ggml_op_name(enum ggml_op op);
GGML_API const char* ggml_unary_op_name(enum ggml_unary_op op);
GGML_API const char* ggml_op_desc(const struct ggml_tensor* t);
GGML_API const char* ggml_status_to_string(enum ggml_status status);
GGML_API bool ggml_is_quantized(enum ggml_type type);
GGML_API bool ggml_is_contiguous(const struct ggml_tensor* tensor);
GGML_API bool ggml_is_transposed(const struct ggml_tensor* tensor);
GGML_API bool ggml_is_permuted(const struct ggml_tensor* tensor);
GGML_API bool ggml_are_same_shape(const struct ggml_tensor* t0, const struct ggml_tensor* t1);
GGML_API bool ggml_can_repeat(const struct ggml_tensor* t0, const struct ggml_tensor* t1);
GGML_API int ggml_n_dims(const struct ggml_tensor* tensor);
GGML_API enum ggml_unary_op ggml_get_unary_op(const struct ggml_tensor* tensor);

/* ---- fp16 / bf16 / quant blocks (ggml-quants) ---- */
GGML_API float ggml_fp16_to_fp32(ggml_fp16_t x);
GGML_API ggml_fp16_t ggml_fp32_to_fp16(float x);
GGML_API void ggml_fp16_to_fp32_row(const ggml_fp16_t* x, float* y, int64_t n);
GGML_API void ggml_fp32_to_fp16_row(const float* x, ggml_fp16_t* y, int64_t n);
GGML_API float ggml_bf16_to_fp32(ggml_bf16_t x);
GGML_API ggml_bf16_t ggml_fp32_to_bf16(float x);
GGML_API void ggml_fp32_to_bf16_row(const float* x, ggml_bf16_t* y, int64_t n);
GGML_API void ggml_bf16_to_fp32_row(const ggml_bf16_t* x, float* y, int64_t n);
/* quantize nrows rows of n_per_row f32 into `type` blocks; returns bytes written (model_loader.cpp:168-202) */
GGML_API size_t ggml_quantize_chunk(enum ggml_type type, const float* src, void* dst, int64_t start, int64_t nrows, int64_t n_per_row, const float* imatrix);
/* dequantize one row (the `to_float` trait) */
GGML_API void ggml_dequantize_row(enum ggml_type type, const void* src, float* dst, int64_t n);

/* ---- op constructors (metadata only) ---- */
GGML_API struct ggml_tensor* ggml_dup(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_add(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_add_inplace(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_sub(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_mul(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_mul_inplace(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_div(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_scale(struct ggml_context* ctx, struct ggml_tensor* a, float s);
GGML_API struct ggml_tensor* ggml_scale_inplace(struct ggml_context* ctx, struct ggml_tensor* a, float s);

GGML_API struct ggml_tensor* ggml_unary(struct ggml_context* ctx, struct ggml_tensor* a, enum ggml_unary_op op);
GGML_API struct ggml_tensor* ggml_unary_inplace(struct ggml_context* ctx, struct ggml_tensor* a, enum ggml_unary_op op);
GGML_API struct ggml_tensor* ggml_silu(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_silu_inplace(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_gelu(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_gelu_inplace(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_gelu_quick(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_gelu_quick_inplace(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_sigmoid(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_tanh(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_relu(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_tanh_inplace(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_relu_inplace(struct ggml_context* ctx, struct ggml_tensor* a);

GGML_API struct ggml_tensor* ggml_norm(struct ggml_context* ctx, struct ggml_tensor* a, float eps);
GGML_API struct ggml_tensor* ggml_rms_norm(struct ggml_context* ctx, struct ggml_tensor* a, float eps);
GGML_API struct ggml_tensor* ggml_group_norm(struct ggml_context* ctx, struct ggml_tensor* a, int n_groups, float eps);

GGML_API struct ggml_tensor* ggml_mul_mat(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API void ggml_mul_mat_set_prec(struct ggml_tensor* a, enum ggml_prec prec);

GGML_API struct ggml_tensor* ggml_cpy(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_cast(struct ggml_context* ctx, struct ggml_tensor* a, enum ggml_type type);
GGML_API struct ggml_tensor* ggml_cont(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_reshape(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_reshape_1d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0);
GGML_API struct ggml_tensor* ggml_reshape_2d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1);
GGML_API struct ggml_tensor* ggml_reshape_3d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2);
GGML_API struct ggml_tensor* ggml_reshape_4d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
GGML_API struct ggml_tensor* ggml_view_1d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, size_t offset);
GGML_API struct ggml_tensor* ggml_view_2d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset);
GGML_API struct ggml_tensor* ggml_view_3d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset);
GGML_API struct ggml_tensor* ggml_view_4d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset);
GGML_API struct ggml_tensor* ggml_permute(struct ggml_context* ctx, struct ggml_tensor* a, int axis0, int axis1, int axis2, int axis3);
GGML_API struct ggml_tensor* ggml_transpose(struct ggml_context* ctx, struct ggml_tensor* a);

GGML_API struct ggml_tensor* ggml_repeat(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);
GGML_API struct ggml_tensor* ggml_repeat_4d(struct ggml_context* ctx, struct ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
GGML_API struct ggml_tensor* ggml_concat(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int dim);
GGML_API struct ggml_tensor* ggml_soft_max(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_soft_max_inplace(struct ggml_context* ctx, struct ggml_tensor* a);
GGML_API struct ggml_tensor* ggml_soft_max_ext(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* mask, float scale, float max_bias);
GGML_API struct ggml_tensor* ggml_get_rows(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b);

GGML_API struct ggml_tensor* ggml_im2col(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1, bool is_2D, enum ggml_type dst_type);
GGML_API struct ggml_tensor* ggml_conv_2d(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1);
GGML_API struct ggml_tensor* ggml_conv_2d_direct(struct ggml_context* ctx, struct ggml_tensor* a, struct ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1);
GGML_API struct ggml_tensor* ggml_upscale(struct ggml_context* ctx, struct ggml_tensor* a, int scale_factor, enum ggml_scale_mode mode);
GGML_API struct ggml_tensor* ggml_pad(struct ggml_context* ctx, struct ggml_tensor* a, int p0, int p1, int p2, int p3);
GGML_API struct ggml_tensor* ggml_pad_ext(struct ggml_context* ctx, struct ggml_tensor* a, int lp0, int rp0, int lp1, int rp1, int lp2, int rp2, int lp3, int rp3);
GGML_API struct ggml_tensor* ggml_timestep_embedding(struct ggml_context* ctx, struct ggml_tensor* timesteps, int dim, int max_period);
GGML_API struct ggml_tensor* ggml_flash_attn_ext(struct ggml_context* ctx, struct ggml_tensor* q, struct ggml_tensor* k, struct ggml_tensor* v, struct ggml_tensor* mask, float scale, float max_bias, float logit_softcap);
GGML_API void ggml_flash_attn_ext_set_prec(struct ggml_tensor* a, enum ggml_prec prec);

/* ---- graphs ---- */
GGML_API struct ggml_cgraph* ggml_new_graph(struct ggml_context* ctx);
GGML_API struct ggml_cgraph* ggml_new_graph_custom(struct ggml_context* ctx, size_t size, bool grads);
GGML_API void ggml_build_forward_expand(struct ggml_cgraph* cgraph, struct ggml_tensor* tensor);
GGML_API int ggml_graph_n_nodes(struct ggml_cgraph* cgraph);
GGML_API struct ggml_tensor* ggml_graph_node(struct ggml_cgraph* cgraph, int i);
GGML_API void ggml_graph_add_node(struct ggml_cgraph* cgraph, struct ggml_tensor* tensor);
GGML_API void ggml_unravel_index(const struct ggml_tensor* tensor, int64_t i, int64_t* i0, int64_t* i1, int64_t* i2, int64_t* i3);
GGML_API int64_t ggml_time_ms(void);
GGML_API int64_t ggml_time_us(void);
GGML_API struct ggml_tensor* ggml_graph_get_tensor(const struct ggml_cgraph* cgraph, const char* name);

/* ---- graph allocator (ggml-alloc.h; reference: ggml_extend.hpp:2227-2232,2832) ---- */
typedef struct ggml_gallocr* ggml_gallocr_t;
GGML_API ggml_gallocr_t ggml_gallocr_new(ggml_backend_buffer_type_t buft);
GGML_API void ggml_gallocr_free(ggml_gallocr_t galloc);
GGML_API bool ggml_gallocr_reserve(ggml_gallocr_t galloc, struct ggml_cgraph* graph);
GGML_API bool ggml_gallocr_alloc_graph(ggml_gallocr_t galloc, struct ggml_cgraph* graph);
GGML_API size_t ggml_gallocr_get_buffer_size(ggml_gallocr_t galloc, int buffer_id);
GGML_API ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors_from_buft(struct ggml_context* ctx, ggml_backend_buffer_type_t buft);
GGML_API ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors(struct ggml_context* ctx, ggml_backend_t backend);
GGML_API enum ggml_status ggml_backend_tensor_alloc(ggml_backend_buffer_t buffer, struct ggml_tensor* tensor, void* addr);

/* ---- backend registry / device / buffer / stream API (ggml-backend.h) ---- */
GGML_API ggml_backend_reg_t ggml_backend_load(const char* path); /* dlopen + ggml_backend_init (GGML_BACKEND_DL) */
GGML_API void ggml_backend_register(ggml_backend_reg_t reg);
GGML_API size_t ggml_backend_reg_count(void);
GGML_API ggml_backend_reg_t ggml_backend_reg_get(size_t index);
GGML_API ggml_backend_reg_t ggml_backend_reg_by_name(const char* name);
GGML_API const char* ggml_backend_reg_name(ggml_backend_reg_t reg);
GGML_API size_t ggml_backend_reg_dev_count(ggml_backend_reg_t reg);
GGML_API ggml_backend_dev_t ggml_backend_reg_dev_get(ggml_backend_reg_t reg, size_t index);
GGML_API void* ggml_backend_reg_get_proc_address(ggml_backend_reg_t reg, const char* name);
GGML_API size_t ggml_backend_dev_count(void);
GGML_API ggml_backend_dev_t ggml_backend_dev_get(size_t index);
GGML_API ggml_backend_dev_t ggml_backend_dev_by_name(const char* name);
GGML_API ggml_backend_dev_t ggml_backend_dev_by_type(enum ggml_backend_dev_type type);
GGML_API const char* ggml_backend_dev_name(ggml_backend_dev_t device);
GGML_API const char* ggml_backend_dev_description(ggml_backend_dev_t device);
GGML_API void ggml_backend_dev_memory(ggml_backend_dev_t device, size_t* free, size_t* total);
GGML_API enum ggml_backend_dev_type ggml_backend_dev_type(ggml_backend_dev_t device);
GGML_API void ggml_backend_dev_get_props(ggml_backend_dev_t device, struct ggml_backend_dev_props* props);
GGML_API ggml_backend_reg_t ggml_backend_dev_backend_reg(ggml_backend_dev_t device);
GGML_API ggml_backend_t ggml_backend_dev_init(ggml_backend_dev_t device, const char* params);
GGML_API ggml_backend_buffer_type_t ggml_backend_dev_buffer_type(ggml_backend_dev_t device);
GGML_API bool ggml_backend_dev_supports_op(ggml_backend_dev_t device, const struct ggml_tensor* op);
GGML_API bool ggml_backend_dev_supports_buft(ggml_backend_dev_t device, ggml_backend_buffer_type_t buft);
GGML_API ggml_backend_t ggml_backend_init_by_name(const char* name, const char* params);
GGML_API ggml_backend_t ggml_backend_init_by_type(enum ggml_backend_dev_type type, const char* params);
GGML_API ggml_backend_t ggml_backend_init_best(void); /* first GPU device, else the first device */
GGML_API void ggml_backend_load_all(void);            /* plug-ins are loaded explicitly (ggml_backend_load): nothing to scan */
GGML_API ggml_backend_dev_t ggml_backend_buft_get_device(ggml_backend_buffer_type_t buft);
GGML_API ggml_backend_buffer_type_t ggml_backend_dev_host_buffer_type(ggml_backend_dev_t device); /* NULL when the device has none */

GGML_API const char* ggml_backend_name(ggml_backend_t backend);
GGML_API void ggml_backend_free(ggml_backend_t backend);
GGML_API ggml_backend_dev_t ggml_backend_get_device(ggml_backend_t backend);
GGML_API ggml_backend_buffer_type_t ggml_backend_get_default_buffer_type(ggml_backend_t backend);
GGML_API ggml_backend_buffer_t ggml_backend_alloc_buffer(ggml_backend_t backend, size_t size);
GGML_API size_t ggml_backend_get_alignment(ggml_backend_t backend);
GGML_API bool ggml_backend_supports_op(ggml_backend_t backend, const struct ggml_tensor* op);
GGML_API void ggml_backend_synchronize(ggml_backend_t backend);
GGML_API enum ggml_status ggml_backend_graph_compute(ggml_backend_t backend, struct ggml_cgraph* cgraph);
GGML_API enum ggml_status ggml_backend_graph_compute_async(ggml_backend_t backend, struct ggml_cgraph* cgraph);

GGML_API const char* ggml_backend_buft_name(ggml_backend_buffer_type_t buft);
GGML_API ggml_backend_buffer_t ggml_backend_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size);
GGML_API size_t ggml_backend_buft_get_alignment(ggml_backend_buffer_type_t buft);
GGML_API size_t ggml_backend_buft_get_max_size(ggml_backend_buffer_type_t buft);
GGML_API size_t ggml_backend_buft_get_alloc_size(ggml_backend_buffer_type_t buft, const struct ggml_tensor* tensor);
GGML_API bool ggml_backend_buft_is_host(ggml_backend_buffer_type_t buft);

GGML_API void ggml_backend_buffer_free(ggml_backend_buffer_t buffer);
GGML_API void* ggml_backend_buffer_get_base(ggml_backend_buffer_t buffer);
GGML_API size_t ggml_backend_buffer_get_size(ggml_backend_buffer_t buffer);
GGML_API void ggml_backend_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value);
GGML_API void ggml_backend_buffer_set_usage(ggml_backend_buffer_t buffer, enum ggml_backend_buffer_usage usage);
GGML_API enum ggml_backend_buffer_usage ggml_backend_buffer_get_usage(ggml_backend_buffer_t buffer);
GGML_API ggml_backend_buffer_type_t ggml_backend_buffer_get_type(ggml_backend_buffer_t buffer);
GGML_API bool ggml_backend_buffer_is_host(ggml_backend_buffer_t buffer);

GGML_API void ggml_backend_tensor_set(struct ggml_tensor* tensor, const void* data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_get(const struct ggml_tensor* tensor, void* data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_set_async(ggml_backend_t backend, struct ggml_tensor* tensor, const void* data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_get_async(ggml_backend_t backend, const struct ggml_tensor* tensor, void* data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_memset(struct ggml_tensor* tensor, uint8_t value, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_copy(struct ggml_tensor* src, struct ggml_tensor* dst);

#define GGML_ASSERT(x)                                                                 \
    do {                                                                               \
        if (!(x)) ggml_abort(__FILE__, __LINE__, "GGML_ASSERT(%s) failed", #x);        \
    } while (0)
GGML_API void ggml_abort(const char* file, int line, const char* fmt, ...) __attribute__((noreturn));
#define GGML_ABORT(...) ggml_abort(__FILE__, __LINE__, __VA_ARGS__)

#ifdef __cplusplus
}
#endif
