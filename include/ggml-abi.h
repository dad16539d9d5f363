/*
 * ggml-abi.h — clean-room declaration of the slice of the ggml ABI that stable-diffusion.cpp's
 * hot path (UNet / MMDiT denoise step + KL-VAE decode) crosses.
 *
 * WHY THIS FILE EXISTS
 *   The reference's `ggml/` directory is an empty, un-vendored git submodule
 *   (/root/reference/.gitmodules:1-3).  The host code above the boundary
 *   (src/core/ggml_extend.hpp, src/core/ggml_extend_backend.cpp) talks to the device ONLY through
 *   ggml's public API and its backend plug-in vtables.  This header re-declares exactly that
 *   surface from upstream knowledge, so that
 *     (1) our MI355X backend (libggml-mi355x.so) is written against the operator interface the
 *         reference host expects (ggml_backend_i / ggml_backend_device_i / ggml_backend_buffer_i), and
 *     (2) everything ABI-uncertain (struct field order, enum numeric values) lives in ONE file that
 *         can be swapped for the real pinned ggml.h / ggml-backend-impl.h when a checkout exists.
 *   The pinned ggml commit is NOT recoverable from the reference tree, so enum numeric values
 *   below are UNPINNED: they follow upstream ggml-org/ggml master order as of mid-2025.
 *
 * Evidence for layout choices (reference file:line):
 *   GGML_MAX_NAME = 160                    CMakeLists.txt:316
 *   ggml_cgraph field order (…, uid)       src/core/ggml_extend_backend.cpp:449-463
 *   sd_type_t "same as enum ggml_type"     include/stable-diffusion.h:98-143
 *   backend vtables                        SURVEY.md Appendix C (recollection of ggml-backend-impl.h)
 */
#ifndef GGML_ABI_H
#define GGML_ABI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MAX_DIMS 4
#define GGML_MAX_SRC 10
#define GGML_MAX_OP_PARAMS 64
#define GGML_MAX_NAME 160 /* reference CMakeLists.txt:316 forces 160 (upstream default 64) */
#define GGML_DEFAULT_GRAPH_SIZE 2048
#define GGML_BACKEND_API_VERSION 2

typedef uint16_t ggml_fp16_t;
typedef struct {
    uint16_t bits;
} ggml_bf16_t;

enum ggml_status {
    GGML_STATUS_ALLOC_FAILED = -2,
    GGML_STATUS_FAILED       = -1,
    GGML_STATUS_SUCCESS      = 0,
    GGML_STATUS_ABORTED      = 1,
};

/* numeric values mirrored by sd_type_t (include/stable-diffusion.h:99-143) */
enum ggml_type {
    GGML_TYPE_F32     = 0,
    GGML_TYPE_F16     = 1,
    GGML_TYPE_Q4_0    = 2,
    GGML_TYPE_Q4_1    = 3,
    GGML_TYPE_Q5_0    = 6,
    GGML_TYPE_Q5_1    = 7,
    GGML_TYPE_Q8_0    = 8,
    GGML_TYPE_Q8_1    = 9,
    GGML_TYPE_Q2_K    = 10,
    GGML_TYPE_Q3_K    = 11,
    GGML_TYPE_Q4_K    = 12,
    GGML_TYPE_Q5_K    = 13,
    GGML_TYPE_Q6_K    = 14,
    GGML_TYPE_Q8_K    = 15,
    GGML_TYPE_IQ2_XXS = 16,
    GGML_TYPE_IQ2_XS  = 17,
    GGML_TYPE_IQ3_XXS = 18,
    GGML_TYPE_IQ1_S   = 19,
    GGML_TYPE_IQ4_NL  = 20,
    GGML_TYPE_IQ3_S   = 21,
    GGML_TYPE_IQ2_S   = 22,
    GGML_TYPE_IQ4_XS  = 23,
    GGML_TYPE_I8      = 24,
    GGML_TYPE_I16     = 25,
    GGML_TYPE_I32     = 26,
    GGML_TYPE_I64     = 27,
    GGML_TYPE_F64     = 28,
    GGML_TYPE_IQ1_M   = 29,
    GGML_TYPE_BF16    = 30,
    GGML_TYPE_TQ1_0   = 34,
    GGML_TYPE_TQ2_0   = 35,
    GGML_TYPE_MXFP4   = 39,
    GGML_TYPE_NVFP4   = 40,
    GGML_TYPE_Q1_0    = 41,
    GGML_TYPE_COUNT   = 42,
};

enum ggml_prec {
    GGML_PREC_DEFAULT = 0,
    GGML_PREC_F32     = 10,
};

/* op order follows upstream ggml.h; numeric values UNPINNED (the leejet fork adds ops) */
enum ggml_op {
    GGML_OP_NONE = 0,
    GGML_OP_DUP,
    GGML_OP_ADD,
    GGML_OP_ADD_ID,
    GGML_OP_ADD1,
#ifdef GGML_ABI_TEST_SHIFTED_ENUMS /* TESTS ONLY: a host "fork" with two ops inserted mid-enum (tests/test_abi.py::test_op_enum_remap); never defined for the plug-in */
    GGML_OP_FORK_EXTRA_A,
    GGML_OP_FORK_EXTRA_B,
#endif
    GGML_OP_ACC,
    GGML_OP_SUB,
    GGML_OP_MUL,
    GGML_OP_DIV,
    GGML_OP_SQR,
    GGML_OP_SQRT,
    GGML_OP_LOG,
    GGML_OP_SIN,
    GGML_OP_COS,
    GGML_OP_SUM,
    GGML_OP_SUM_ROWS,
    GGML_OP_CUMSUM,
    GGML_OP_MEAN,
    GGML_OP_ARGMAX,
    GGML_OP_COUNT_EQUAL,
    GGML_OP_REPEAT,
    GGML_OP_REPEAT_BACK,
    GGML_OP_CONCAT,
    GGML_OP_SILU_BACK,
    GGML_OP_NORM,
    GGML_OP_RMS_NORM,
    GGML_OP_RMS_NORM_BACK,
    GGML_OP_GROUP_NORM,
    GGML_OP_L2_NORM,
    GGML_OP_MUL_MAT,
    GGML_OP_MUL_MAT_ID,
    GGML_OP_OUT_PROD,
    GGML_OP_SCALE,
    GGML_OP_SET,
    GGML_OP_CPY,
    GGML_OP_CONT,
    GGML_OP_RESHAPE,
    GGML_OP_VIEW,
    GGML_OP_PERMUTE,
    GGML_OP_TRANSPOSE,
    GGML_OP_GET_ROWS,
    GGML_OP_GET_ROWS_BACK,
    GGML_OP_SET_ROWS,
    GGML_OP_DIAG,
    GGML_OP_DIAG_MASK_INF,
    GGML_OP_DIAG_MASK_ZERO,
    GGML_OP_SOFT_MAX,
    GGML_OP_SOFT_MAX_BACK,
    GGML_OP_ROPE,
    GGML_OP_ROPE_BACK,
    GGML_OP_CLAMP,
    GGML_OP_CONV_TRANSPOSE_1D,
    GGML_OP_IM2COL,
    GGML_OP_IM2COL_BACK,
    GGML_OP_IM2COL_3D,
    GGML_OP_CONV_2D,
    GGML_OP_CONV_3D,
    GGML_OP_CONV_2D_DW,
    GGML_OP_CONV_TRANSPOSE_2D,
    GGML_OP_POOL_1D,
    GGML_OP_POOL_2D,
    GGML_OP_POOL_2D_BACK,
    GGML_OP_UPSCALE,
    GGML_OP_PAD,
    GGML_OP_PAD_REFLECT_1D,
    GGML_OP_ROLL,
    GGML_OP_ARANGE,
    GGML_OP_TIMESTEP_EMBEDDING,
    GGML_OP_ARGSORT,
    GGML_OP_TOP_K,
    GGML_OP_LEAKY_RELU,
    GGML_OP_TRI,
    GGML_OP_FILL,
    GGML_OP_FLASH_ATTN_EXT,
    GGML_OP_FLASH_ATTN_BACK,
    GGML_OP_SSM_CONV,
    GGML_OP_SSM_SCAN,
    GGML_OP_WIN_PART,
    GGML_OP_WIN_UNPART,
    GGML_OP_GET_REL_POS,
    GGML_OP_ADD_REL_POS,
    GGML_OP_RWKV_WKV6,
    GGML_OP_GATED_LINEAR_ATTN,
    GGML_OP_RWKV_WKV7,
    GGML_OP_SOLVE_TRI,
    GGML_OP_UNARY,
    GGML_OP_MAP_CUSTOM1,
    GGML_OP_MAP_CUSTOM2,
    GGML_OP_MAP_CUSTOM3,
    GGML_OP_CUSTOM,
    GGML_OP_CROSS_ENTROPY_LOSS,
    GGML_OP_CROSS_ENTROPY_LOSS_BACK,
    GGML_OP_OPT_STEP_ADAMW,
    GGML_OP_OPT_STEP_SGD,
    GGML_OP_GLU,
    GGML_OP_COUNT,
};

enum ggml_unary_op {
    GGML_UNARY_OP_ABS = 0,
    GGML_UNARY_OP_SGN,
    GGML_UNARY_OP_NEG,
    GGML_UNARY_OP_STEP,
#ifdef GGML_ABI_TEST_SHIFTED_ENUMS
    GGML_UNARY_OP_FORK_EXTRA,
#endif
    GGML_UNARY_OP_TANH,
    GGML_UNARY_OP_ELU,
    GGML_UNARY_OP_RELU,
    GGML_UNARY_OP_SIGMOID,
    GGML_UNARY_OP_GELU,
    GGML_UNARY_OP_GELU_QUICK,
    GGML_UNARY_OP_SILU,
    GGML_UNARY_OP_HARDSWISH,
    GGML_UNARY_OP_HARDSIGMOID,
    GGML_UNARY_OP_EXP,
    GGML_UNARY_OP_GELU_ERF,
    GGML_UNARY_OP_COUNT,
};

enum ggml_scale_mode {
    GGML_SCALE_MODE_NEAREST  = 0,
    GGML_SCALE_MODE_BILINEAR = 1,
    GGML_SCALE_MODE_BICUBIC  = 2,
    GGML_SCALE_MODE_COUNT,
};

enum ggml_tensor_flag {
    GGML_TENSOR_FLAG_INPUT  = 1,
    GGML_TENSOR_FLAG_OUTPUT = 2,
    GGML_TENSOR_FLAG_PARAM  = 4,
    GGML_TENSOR_FLAG_LOSS   = 8,
};

struct ggml_backend_buffer;
struct ggml_context;

/* n-dimensional tensor; ne[0] is the contiguous ("fastest") dimension */
struct ggml_tensor {
    enum ggml_type type;
    struct ggml_backend_buffer* buffer;
    int64_t ne[GGML_MAX_DIMS]; /* number of elements */
    size_t nb[GGML_MAX_DIMS];  /* stride in bytes: nb[0]=type_size, nb[i]=nb[i-1]*ne[i-1] (+padding) */
    enum ggml_op op;
    int32_t op_params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];
    int32_t flags;
    struct ggml_tensor* src[GGML_MAX_SRC];
    struct ggml_tensor* view_src; /* source tensor for views */
    size_t view_offs;             /* offset within view_src->data */
    void* data;                   /* device pointer inside `buffer` once allocated */
    char name[GGML_MAX_NAME];
    void* extra;
    char padding[8];
};

enum ggml_cgraph_eval_order {
    GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT = 0,
    GGML_CGRAPH_EVAL_ORDER_RIGHT_TO_LEFT,
    GGML_CGRAPH_EVAL_ORDER_COUNT
};

struct ggml_hash_set {
    size_t size;
    uint32_t* used; /* bitset */
    struct ggml_tensor** keys;
};

/* field order evidenced by src/core/ggml_extend_backend.cpp:449-463 (sub-graph view construction) */
struct ggml_cgraph {
    int size;
    int n_nodes;
    int n_leafs;
    struct ggml_tensor** nodes;     /* tensors with op != NONE, topological order */
    struct ggml_tensor** grads;     /* unused (inference) */
    struct ggml_tensor** grad_accs; /* unused (inference) */
    struct ggml_tensor** leafs;     /* may be NULL on sub-graph views */
    int32_t* use_counts;
    struct ggml_hash_set visited_hash_set;
    enum ggml_cgraph_eval_order order;
    uint64_t uid;
};

/* ---------------------------------------------------------------------------------------------
 * backend plug-in interface (ggml-backend-impl.h, API version 2) — the DROP-IN BOUNDARY.
 * Reference call sites: src/core/ggml_extend_backend.cpp:302-320,393-418,466-509;
 *                       src/core/ggml_extend.hpp:2212-2245,2347-2435,2832-2865
 * ------------------------------------------------------------------------------------------- */
typedef struct ggml_backend_buffer_type* ggml_backend_buffer_type_t;
typedef struct ggml_backend_buffer* ggml_backend_buffer_t;
typedef struct ggml_backend_event* ggml_backend_event_t;
typedef struct ggml_backend* ggml_backend_t;
typedef struct ggml_backend_reg* ggml_backend_reg_t;
typedef struct ggml_backend_device* ggml_backend_dev_t;
typedef void* ggml_backend_graph_plan_t;

enum ggml_backend_buffer_usage {
    GGML_BACKEND_BUFFER_USAGE_ANY     = 0,
    GGML_BACKEND_BUFFER_USAGE_WEIGHTS = 1,
    GGML_BACKEND_BUFFER_USAGE_COMPUTE = 2,
};

enum ggml_backend_dev_type {
    GGML_BACKEND_DEVICE_TYPE_CPU,
    GGML_BACKEND_DEVICE_TYPE_GPU,
    GGML_BACKEND_DEVICE_TYPE_IGPU,
    GGML_BACKEND_DEVICE_TYPE_ACCEL
};

struct ggml_backend_dev_caps {
    bool async;
    bool host_buffer;
    bool buffer_from_host_ptr;
    bool events;
};

struct ggml_backend_dev_props {
    const char* name;
    const char* description;
    size_t memory_free;
    size_t memory_total;
    enum ggml_backend_dev_type type;
    const char* device_id;
    struct ggml_backend_dev_caps caps;
};

struct ggml_backend_buffer_type_i {
    const char* (*get_name)(ggml_backend_buffer_type_t buft);
    ggml_backend_buffer_t (*alloc_buffer)(ggml_backend_buffer_type_t buft, size_t size);
    size_t (*get_alignment)(ggml_backend_buffer_type_t buft);
    size_t (*get_max_size)(ggml_backend_buffer_type_t buft);
    size_t (*get_alloc_size)(ggml_backend_buffer_type_t buft, const struct ggml_tensor* tensor);
    bool (*is_host)(ggml_backend_buffer_type_t buft);
};

struct ggml_backend_buffer_type {
    struct ggml_backend_buffer_type_i iface;
    ggml_backend_dev_t device;
    void* context;
};

struct ggml_backend_buffer_i {
    void (*free_buffer)(ggml_backend_buffer_t buffer);
    void* (*get_base)(ggml_backend_buffer_t buffer);
    enum ggml_status (*init_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor* tensor);
    void (*memset_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor* tensor, uint8_t value, size_t offset, size_t size);
    void (*set_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor* tensor, const void* data, size_t offset, size_t size);
    void (*get_tensor)(ggml_backend_buffer_t buffer, const struct ggml_tensor* tensor, void* data, size_t offset, size_t size);
    bool (*cpy_tensor)(ggml_backend_buffer_t buffer, const struct ggml_tensor* src, struct ggml_tensor* dst);
    void (*clear)(ggml_backend_buffer_t buffer, uint8_t value);
    void (*reset)(ggml_backend_buffer_t buffer);
};

struct ggml_backend_buffer {
    struct ggml_backend_buffer_i iface;
    ggml_backend_buffer_type_t buft;
    void* context;
    size_t size;
    enum ggml_backend_buffer_usage usage;
};

struct ggml_backend_i {
    const char* (*get_name)(ggml_backend_t backend);
    void (*free)(ggml_backend_t backend);
    void (*set_tensor_async)(ggml_backend_t backend, struct ggml_tensor* tensor, const void* data, size_t offset, size_t size);
    void (*get_tensor_async)(ggml_backend_t backend, const struct ggml_tensor* tensor, void* data, size_t offset, size_t size);
    bool (*cpy_tensor_async)(ggml_backend_t backend_src, ggml_backend_t backend_dst, const struct ggml_tensor* src, struct ggml_tensor* dst);
    void (*synchronize)(ggml_backend_t backend);
    ggml_backend_graph_plan_t (*graph_plan_create)(ggml_backend_t backend, const struct ggml_cgraph* cgraph);
    void (*graph_plan_free)(ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    void (*graph_plan_update)(ggml_backend_t backend, ggml_backend_graph_plan_t plan, const struct ggml_cgraph* cgraph);
    enum ggml_status (*graph_plan_compute)(ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    /* THE hot-path entry: src/core/ggml_extend_backend.cpp:471 (and :489 on sub-graph views) */
    enum ggml_status (*graph_compute)(ggml_backend_t backend, struct ggml_cgraph* cgraph);
    void (*event_record)(ggml_backend_t backend, ggml_backend_event_t event);
    void (*event_wait)(ggml_backend_t backend, ggml_backend_event_t event);
    void (*graph_optimize)(ggml_backend_t backend, struct ggml_cgraph* cgraph);
};

typedef struct {
    uint8_t b[16];
} ggml_guid;
typedef ggml_guid* ggml_guid_t;

struct ggml_backend {
    ggml_guid_t guid;
    struct ggml_backend_i iface;
    ggml_backend_dev_t device;
    void* context;
};

struct ggml_backend_device_i {
    const char* (*get_name)(ggml_backend_dev_t dev);
    const char* (*get_description)(ggml_backend_dev_t dev);
    void (*get_memory)(ggml_backend_dev_t dev, size_t* free, size_t* total);
    enum ggml_backend_dev_type (*get_type)(ggml_backend_dev_t dev);
    void (*get_props)(ggml_backend_dev_t dev, struct ggml_backend_dev_props* props);
    ggml_backend_t (*init_backend)(ggml_backend_dev_t dev, const char* params);
    ggml_backend_buffer_type_t (*get_buffer_type)(ggml_backend_dev_t dev);
    ggml_backend_buffer_type_t (*get_host_buffer_type)(ggml_backend_dev_t dev);
    ggml_backend_buffer_t (*buffer_from_host_ptr)(ggml_backend_dev_t dev, void* ptr, size_t size, size_t max_tensor_size);
    /* probed at graph-BUILD time: src/core/ggml_extend.hpp:1425 (flash-attn), :2198-2210 (whole graph) */
    bool (*supports_op)(ggml_backend_dev_t dev, const struct ggml_tensor* op);
    bool (*supports_buft)(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft);
    bool (*offload_op)(ggml_backend_dev_t dev, const struct ggml_tensor* op);
    ggml_backend_event_t (*event_new)(ggml_backend_dev_t dev);
    void (*event_free)(ggml_backend_dev_t dev, ggml_backend_event_t event);
    void (*event_synchronize)(ggml_backend_dev_t dev, ggml_backend_event_t event);
};

struct ggml_backend_device {
    struct ggml_backend_device_i iface;
    ggml_backend_reg_t reg;
    void* context;
};

struct ggml_backend_reg_i {
    const char* (*get_name)(ggml_backend_reg_t reg);
    size_t (*get_device_count)(ggml_backend_reg_t reg);
    ggml_backend_dev_t (*get_device)(ggml_backend_reg_t reg, size_t index);
    void* (*get_proc_address)(ggml_backend_reg_t reg, const char* name);
};

struct ggml_backend_reg {
    int api_version; /* GGML_BACKEND_API_VERSION */
    struct ggml_backend_reg_i iface;
    void* context;
};

/* dynamic-loading entry points a libggml-<name>.so exports (GGML_BACKEND_DL;
 * reference: src/core/ggml_extend_backend.cpp:302-320, .github/workflows/build.yml:88,686) */
typedef ggml_backend_reg_t (*ggml_backend_init_t)(void);
typedef int (*ggml_backend_score_t)(void);

/* ---------------------------------------------------------------------------------------------
 * small inline helpers every side needs (kept here so a backend .so has NO link dependency on a
 * ggml-base library)
 * ------------------------------------------------------------------------------------------- */
static inline int64_t ggml_abi_blck_size(enum ggml_type t) {
    switch (t) {
        case GGML_TYPE_Q4_0:
        case GGML_TYPE_Q4_1:
        case GGML_TYPE_Q5_0:
        case GGML_TYPE_Q5_1:
        case GGML_TYPE_Q8_0:
        case GGML_TYPE_Q8_1:
            return 32;
        default:
            return 1;
    }
}
static inline size_t ggml_abi_type_size(enum ggml_type t) {
    switch (t) {
        case GGML_TYPE_F32: return 4;
        case GGML_TYPE_F16: return 2;
        case GGML_TYPE_BF16: return 2;
        case GGML_TYPE_Q4_0: return 18; /* half d; uint8 qs[16] */
        case GGML_TYPE_Q4_1: return 20;
        case GGML_TYPE_Q5_0: return 22;
        case GGML_TYPE_Q5_1: return 24;
        case GGML_TYPE_Q8_0: return 34; /* half d; int8 qs[32] */
        case GGML_TYPE_Q8_1: return 36;
        case GGML_TYPE_I8: return 1;
        case GGML_TYPE_I16: return 2;
        case GGML_TYPE_I32: return 4;
        case GGML_TYPE_I64: return 8;
        case GGML_TYPE_F64: return 8;
        default: return 0;
    }
}
static inline size_t ggml_abi_row_size(enum ggml_type t, int64_t ne) {
    return ggml_abi_type_size(t) * (size_t)(ne / ggml_abi_blck_size(t));
}
static inline int64_t ggml_abi_nelements(const struct ggml_tensor* t) {
    return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3];
}
static inline int64_t ggml_abi_nrows(const struct ggml_tensor* t) {
    return t->ne[1] * t->ne[2] * t->ne[3];
}
static inline size_t ggml_abi_nbytes(const struct ggml_tensor* t) {
    for (int i = 0; i < GGML_MAX_DIMS; ++i) {
        if (t->ne[i] <= 0) return 0;
    }
    const int64_t blck = ggml_abi_blck_size(t->type);
    size_t nbytes;
    if (blck == 1) {
        nbytes = ggml_abi_type_size(t->type);
        for (int i = 0; i < GGML_MAX_DIMS; ++i) nbytes += (size_t)(t->ne[i] - 1) * t->nb[i];
    } else {
        nbytes = (size_t)t->ne[0] * t->nb[0] / (size_t)blck;
        for (int i = 1; i < GGML_MAX_DIMS; ++i) nbytes += (size_t)(t->ne[i] - 1) * t->nb[i];
    }
    return nbytes;
}
static inline bool ggml_abi_is_contiguous(const struct ggml_tensor* t) {
    size_t next = ggml_abi_type_size(t->type);
    const int64_t blck = ggml_abi_blck_size(t->type);
    if (t->ne[0] != blck && t->nb[0] != next) return false;
    next *= (size_t)(t->ne[0] / blck);
    for (int i = 1; i < GGML_MAX_DIMS; ++i) {
        if (t->ne[i] != 1) {
            if (t->nb[i] != next) return false;
            next *= (size_t)t->ne[i];
        }
    }
    return true;
}
static inline bool ggml_abi_same_shape(const struct ggml_tensor* a, const struct ggml_tensor* b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}
static inline enum ggml_unary_op ggml_abi_get_unary_op(const struct ggml_tensor* t) {
    return (enum ggml_unary_op)t->op_params[0];
}
static inline float ggml_abi_op_param_f32(const struct ggml_tensor* t, int i) {
    union {
        int32_t i;
        float f;
    } u;
    u.i = t->op_params[i];
    return u.f;
}
/* ops that carry no work (must be no-ops in graph_compute; SURVEY.md §2.3 last rows) */
static inline bool ggml_abi_op_is_noop(enum ggml_op op) {
    return op == GGML_OP_NONE || op == GGML_OP_RESHAPE || op == GGML_OP_VIEW || op == GGML_OP_PERMUTE ||
           op == GGML_OP_TRANSPOSE;
}

#ifdef __cplusplus
}
#endif
#endif /* GGML_ABI_H */
