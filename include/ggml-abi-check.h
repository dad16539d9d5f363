/*
 * ggml-abi-check.h — compile-time pin of every layout fact of ggml's headers that libggml-mi355x.so depends on.
 *
 * include/ggml-abi.h is a clean-room restatement of upstream ggml.h / ggml-backend-impl.h (the submodule is empty in the reference
 * drop), so it cannot be validated here.  This header states what the backend ASSUMES, as static_asserts on whatever
 * declarations are in scope: include it after ggml-abi.h (as csrc/backend/backend.cpp does) or — when a real ggml checkout is
 * mounted and the backend is rebuilt against it — after the real ggml.h + ggml-backend-impl.h.  A disagreement then stops the build
 * instead of corrupting memory at run time.  Values: GGML_MAX_NAME = 160 (reference CMakeLists.txt:316), LP64.
 */
#ifndef GGML_ABI_CHECK_H
#define GGML_ABI_CHECK_H
#include <stddef.h>

#ifdef __cplusplus
#define GGML_ABI_ASSERT(c, m) static_assert(c, m)
#else
#define GGML_ABI_ASSERT(c, m) _Static_assert(c, m)
#endif

GGML_ABI_ASSERT(GGML_MAX_DIMS == 4 && GGML_MAX_SRC == 10 && GGML_MAX_OP_PARAMS == 64, "ggml tensor limits changed");
GGML_ABI_ASSERT(GGML_MAX_NAME == 160, "stable-diffusion.cpp builds ggml with GGML_MAX_NAME=160 (CMakeLists.txt:316)");

/* struct ggml_tensor: the planner reads type, ne, nb, op, op_params, flags, src, view_src, view_offs, data, name */
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, type) == 0, "ggml_tensor.type");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, buffer) == 8, "ggml_tensor.buffer");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, ne) == 16, "ggml_tensor.ne");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, nb) == 48, "ggml_tensor.nb");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, op) == 80, "ggml_tensor.op");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, op_params) == 84, "ggml_tensor.op_params");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, flags) == 148, "ggml_tensor.flags");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, src) == 152, "ggml_tensor.src");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, view_src) == 232, "ggml_tensor.view_src");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, view_offs) == 240, "ggml_tensor.view_offs");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, data) == 248, "ggml_tensor.data");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, name) == 256, "ggml_tensor.name");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, extra) == 416, "ggml_tensor.extra");
GGML_ABI_ASSERT(sizeof(struct ggml_tensor) == 432, "sizeof(ggml_tensor)");

/* struct ggml_cgraph: field ORDER as the reference's sub-graph view initialiser spells it (src/core/ggml_extend_backend.cpp:449-463):
 * size, n_nodes, n_leafs, nodes, grads, grad_accs, leafs, use_counts, visited_hash_set, order, uid */
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, size) == 0 && offsetof(struct ggml_cgraph, n_nodes) == 4 && offsetof(struct ggml_cgraph, n_leafs) == 8,
                "ggml_cgraph counters");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, nodes) == 16, "ggml_cgraph.nodes");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, grads) == 24 && offsetof(struct ggml_cgraph, grad_accs) == 32, "ggml_cgraph.grads / grad_accs");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, leafs) == 40, "ggml_cgraph.leafs");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, use_counts) == 48, "ggml_cgraph.use_counts");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, visited_hash_set) == 56, "ggml_cgraph.visited_hash_set");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, order) == 80 && offsetof(struct ggml_cgraph, uid) == 88, "ggml_cgraph.order / uid");

/* numeric values the planner switches on and the weight decoders key on: enum ggml_type (= sd_type_t, stable-diffusion.h:99-143) */
GGML_ABI_ASSERT(GGML_TYPE_F32 == 0 && GGML_TYPE_F16 == 1 && GGML_TYPE_Q4_0 == 2 && GGML_TYPE_Q8_0 == 8 && GGML_TYPE_I32 == 26 && GGML_TYPE_BF16 == 30,
                "ggml_type numbering");

/* enum ggml_op / ggml_unary_op: the numbers a PREBUILT libggml-mi355x.so was compiled with (include/ggml-abi.h's recollection of upstream).  The
 * prebuilt plug-in does not trust them at run time — ggml_backend_init() rebuilds its op tables from the host's ggml_op_name() /
 * ggml_unary_op_name() (csrc/backend/backend.cpp: resolve_host_enums; reference call sites src/core/ggml_extend_backend.cpp:302-320, 466-509) —
 * so a fork that inserts ops is translated, not mis-dispatched.  When this header is included after a REAL ggml.h these asserts tell the maintainer
 * whether that translation is the identity for this checkout; define GGML_ABI_CHECK_SKIP_OP_NUMBERS to rebuild against a fork with shifted numbers
 * (the rebuilt plug-in then uses the fork's constants directly). */
#ifndef GGML_ABI_CHECK_SKIP_OP_NUMBERS
GGML_ABI_ASSERT(GGML_OP_NONE == 0 && GGML_OP_DUP == 1 && GGML_OP_ADD == 2 && GGML_OP_SUB == 6 && GGML_OP_MUL == 7 && GGML_OP_DIV == 8, "ggml_op numbering: arithmetic");
GGML_ABI_ASSERT(GGML_OP_REPEAT == 20 && GGML_OP_CONCAT == 22 && GGML_OP_NORM == 24 && GGML_OP_RMS_NORM == 25 && GGML_OP_GROUP_NORM == 27,
                "ggml_op numbering: repeat / concat / norms");
GGML_ABI_ASSERT(GGML_OP_MUL_MAT == 29 && GGML_OP_SCALE == 32 && GGML_OP_CPY == 34 && GGML_OP_CONT == 35 && GGML_OP_RESHAPE == 36 && GGML_OP_VIEW == 37 &&
                    GGML_OP_PERMUTE == 38 && GGML_OP_TRANSPOSE == 39 && GGML_OP_GET_ROWS == 40,
                "ggml_op numbering: mul_mat .. get_rows");
GGML_ABI_ASSERT(GGML_OP_SOFT_MAX == 46 && GGML_OP_IM2COL == 52 && GGML_OP_CONV_2D == 55 && GGML_OP_UPSCALE == 62 && GGML_OP_PAD == 63 &&
                    GGML_OP_TIMESTEP_EMBEDDING == 67 && GGML_OP_FLASH_ATTN_EXT == 73 && GGML_OP_UNARY == 85,
                "ggml_op numbering: soft_max .. unary");
GGML_ABI_ASSERT(GGML_UNARY_OP_NEG == 2 && GGML_UNARY_OP_TANH == 4 && GGML_UNARY_OP_RELU == 6 && GGML_UNARY_OP_SIGMOID == 7 && GGML_UNARY_OP_GELU == 8 &&
                    GGML_UNARY_OP_GELU_QUICK == 9 && GGML_UNARY_OP_SILU == 10 && GGML_UNARY_OP_EXP == 13,
                "ggml_unary_op numbering");
#endif
GGML_ABI_ASSERT(GGML_OP_COUNT < 255 && GGML_UNARY_OP_COUNT < 255, "the by-name translation tables are indexed by a byte");

/* plug-in vtables: the entry points the reference calls must sit where upstream puts them (ggml-backend-impl.h, API version 2) */
GGML_ABI_ASSERT(GGML_BACKEND_API_VERSION == 2, "backend API version");
GGML_ABI_ASSERT(offsetof(struct ggml_backend, iface) == 8, "ggml_backend.iface");
GGML_ABI_ASSERT(offsetof(struct ggml_backend_i, get_name) == 0 && offsetof(struct ggml_backend_i, free) == 8, "ggml_backend_i head");
GGML_ABI_ASSERT(offsetof(struct ggml_backend_i, set_tensor_async) == 16 && offsetof(struct ggml_backend_i, get_tensor_async) == 24 &&
                    offsetof(struct ggml_backend_i, cpy_tensor_async) == 32 && offsetof(struct ggml_backend_i, synchronize) == 40,
                "ggml_backend_i async block");
GGML_ABI_ASSERT(offsetof(struct ggml_backend_i, graph_compute) == 80, "ggml_backend_i.graph_compute (after the four graph_plan_* slots)");
GGML_ABI_ASSERT(offsetof(struct ggml_backend_reg, api_version) == 0 && offsetof(struct ggml_backend_reg, iface) == 8, "ggml_backend_reg head");

#endif
