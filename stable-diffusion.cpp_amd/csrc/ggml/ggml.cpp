// ggml.cpp — mini graph front-end (metadata only): contexts, tensors, op constructors, graphs,
// fp16/bf16 conversion and the Q8_0 / Q4_0 block encoders.  See ggml.h for scope and provenance.
// Semantics restated from upstream ggml (absent from /root/reference — SURVEY.md F1); each op
// constructor documents the result shape / op_params encoding the backends rely on.
#include "ggml.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <ctime>
#include <vector>

extern "C" void ggml_abort(const char* file, int line, const char* fmt, ...) {
    fflush(stdout);
    fprintf(stderr, "%s:%d: ", file, line);
    va_list args;
    va_start(args, fmt);
    vfprintf(stderr, fmt, args);
    va_end(args);
    fprintf(stderr, "\n");
    if (getenv("GGML_ABORT_BACKTRACE")) {  // which constructor / wrapper reached the failed assertion
        void* frames[48];
        backtrace_symbols_fd(frames, backtrace(frames, 48), 2);
    }
    abort();
}

// The visited set is upstream's open-addressing ggml_hash_set (keys + a used bitset, hash = address >> 4, linear probing) and use_counts[slot] counts
// how often the tensor in that slot is a source of a visited tensor: both are part of struct ggml_cgraph and are what a backend handed a SUB-GRAPH VIEW
// (sd_ggml_graph_view, src/core/ggml_extend_backend.cpp:449-463: nodes + i0, leafs = NULL, the PARENT's use_counts / visited_hash_set) has to read to
// tell whether a node of the slice is still needed outside it.
struct graph_storage {
    ggml_cgraph g;  // must be first: the cgraph pointer handed to backends is the storage pointer
    std::vector<ggml_tensor*> nodes, leafs, keys;
    std::vector<uint32_t> used;
    std::vector<int32_t> use_counts;
};

static size_t hash_size_for(size_t min_sz) {  // a prime >= min_sz (upstream picks from a table of primes; any size works for a reader)
    size_t n = min_sz | 1;
    for (;; n += 2) {
        bool prime = n > 2;
        for (size_t d = 3; d * d <= n && prime; d += 2) prime = n % d != 0;
        if (prime) return n;
    }
}
static inline size_t hash_slot(const ggml_hash_set& hs, const ggml_tensor* key) {  // slot holding key, or the free slot where it belongs
    const size_t h = ((size_t)(uintptr_t)key >> 4) % hs.size;
    size_t i       = h;
    while ((hs.used[i >> 5] >> (i & 31) & 1u) && hs.keys[i] != key) {
        i = (i + 1) % hs.size;
        GGML_ASSERT(i != h && "graph hash set full");
    }
    return i;
}
static inline bool hash_has(const ggml_hash_set& hs, size_t slot) { return (hs.used[slot >> 5] >> (slot & 31)) & 1u; }
// true when key was not in the set yet
static inline bool hash_insert(graph_storage* gs, ggml_tensor* key) {
    ggml_hash_set& hs = gs->g.visited_hash_set;
    const size_t i    = hash_slot(hs, key);
    if (hash_has(hs, i)) return false;
    hs.used[i >> 5] |= 1u << (i & 31);
    hs.keys[i]             = key;
    gs->g.use_counts[i]    = 0;
    return true;
}

struct ggml_context {
    bool no_alloc;
    std::vector<ggml_tensor*> tensors;
    std::vector<void*> blobs;             // host tensor data owned by the context
    std::vector<graph_storage*> graphs;   // graphs created in this context
};

extern "C" {

ggml_context* ggml_init(ggml_init_params params) {
    ggml_context* ctx = new ggml_context();
    ctx->no_alloc     = params.no_alloc;
    return ctx;
}

void ggml_free(ggml_context* ctx) {
    if (!ctx) return;
    for (auto* t : ctx->tensors) free(t);
    for (auto* b : ctx->blobs) free(b);
    for (auto* g : ctx->graphs) delete g;
    delete ctx;
}

bool ggml_get_no_alloc(ggml_context* ctx) { return ctx->no_alloc; }
size_t ggml_tensor_overhead(void) { return sizeof(ggml_tensor) + 32; }
size_t ggml_graph_overhead_custom(size_t size, bool) { return sizeof(ggml_cgraph) + size * 4 * sizeof(void*); }
size_t ggml_graph_overhead(void) { return ggml_graph_overhead_custom(GGML_DEFAULT_GRAPH_SIZE, false); }

int64_t ggml_blck_size(enum ggml_type type) { return ggml_abi_blck_size(type); }
size_t ggml_type_size(enum ggml_type type) { return ggml_abi_type_size(type); }
size_t ggml_row_size(enum ggml_type type, int64_t ne) { return ggml_abi_row_size(type, ne); }
int64_t ggml_nelements(const ggml_tensor* t) { return ggml_abi_nelements(t); }
int64_t ggml_nrows(const ggml_tensor* t) { return ggml_abi_nrows(t); }
size_t ggml_nbytes(const ggml_tensor* t) { return ggml_abi_nbytes(t); }
size_t ggml_element_size(const ggml_tensor* t) { return ggml_abi_type_size(t->type); }
bool ggml_is_contiguous(const ggml_tensor* t) { return ggml_abi_is_contiguous(t); }
bool ggml_is_transposed(const ggml_tensor* t) { return t->nb[0] > t->nb[1]; }
bool ggml_is_permuted(const ggml_tensor* t) { return t->nb[0] > t->nb[1] || t->nb[1] > t->nb[2] || t->nb[2] > t->nb[3]; }
bool ggml_are_same_shape(const ggml_tensor* a, const ggml_tensor* b) { return ggml_abi_same_shape(a, b); }
bool ggml_can_repeat(const ggml_tensor* t0, const ggml_tensor* t1) {
    // t0 can be tiled to t1's shape
    for (int i = 0; i < GGML_MAX_DIMS; ++i) {
        if (t0->ne[i] == 0 || t1->ne[i] % t0->ne[i] != 0) return false;
    }
    return true;
}
int ggml_n_dims(const ggml_tensor* t) {
    for (int i = GGML_MAX_DIMS - 1; i >= 1; --i) {
        if (t->ne[i] > 1) return i + 1;
    }
    return 1;
}
bool ggml_is_quantized(enum ggml_type type) { return ggml_abi_blck_size(type) > 1; }
enum ggml_unary_op ggml_get_unary_op(const ggml_tensor* t) {
    GGML_ASSERT(t->op == GGML_OP_UNARY);
    return ggml_abi_get_unary_op(t);
}

const char* ggml_type_name(enum ggml_type type) {
    switch (type) {
        case GGML_TYPE_F32: return "f32";
        case GGML_TYPE_F16: return "f16";
        case GGML_TYPE_BF16: return "bf16";
        case GGML_TYPE_Q4_0: return "q4_0";
        case GGML_TYPE_Q8_0: return "q8_0";
        case GGML_TYPE_I8: return "i8";
        case GGML_TYPE_I16: return "i16";
        case GGML_TYPE_I32: return "i32";
        case GGML_TYPE_I64: return "i64";
        case GGML_TYPE_F64: return "f64";
        default: return "unsupported";
    }
}

// every op of the enum has its name, like upstream's GGML_OP_NAME table (the plug-in resolves the numbers it needs BY NAME and ends its scan at the first
// string that is not an op name, planner.cpp build_map)
const char* ggml_op_name(enum ggml_op op) {
    switch (op) {
        case GGML_OP_NONE: return "NONE";
        case GGML_OP_DUP: return "DUP";
        case GGML_OP_ADD: return "ADD";
        case GGML_OP_ADD_ID: return "ADD_ID";
        case GGML_OP_ADD1: return "ADD1";
#ifdef GGML_ABI_TEST_SHIFTED_ENUMS
        case GGML_OP_FORK_EXTRA_A: return "FORK_EXTRA_A";
        case GGML_OP_FORK_EXTRA_B: return "FORK_EXTRA_B";
#endif
        case GGML_OP_ACC: return "ACC";
        case GGML_OP_SUB: return "SUB";
        case GGML_OP_MUL: return "MUL";
        case GGML_OP_DIV: return "DIV";
        case GGML_OP_SQR: return "SQR";
        case GGML_OP_SQRT: return "SQRT";
        case GGML_OP_LOG: return "LOG";
        case GGML_OP_SIN: return "SIN";
        case GGML_OP_COS: return "COS";
        case GGML_OP_SUM: return "SUM";
        case GGML_OP_SUM_ROWS: return "SUM_ROWS";
        case GGML_OP_CUMSUM: return "CUMSUM";
        case GGML_OP_MEAN: return "MEAN";
        case GGML_OP_ARGMAX: return "ARGMAX";
        case GGML_OP_COUNT_EQUAL: return "COUNT_EQUAL";
        case GGML_OP_REPEAT: return "REPEAT";
        case GGML_OP_REPEAT_BACK: return "REPEAT_BACK";
        case GGML_OP_CONCAT: return "CONCAT";
        case GGML_OP_SILU_BACK: return "SILU_BACK";
        case GGML_OP_NORM: return "NORM";
        case GGML_OP_RMS_NORM: return "RMS_NORM";
        case GGML_OP_RMS_NORM_BACK: return "RMS_NORM_BACK";
        case GGML_OP_GROUP_NORM: return "GROUP_NORM";
        case GGML_OP_L2_NORM: return "L2_NORM";
        case GGML_OP_MUL_MAT: return "MUL_MAT";
        case GGML_OP_MUL_MAT_ID: return "MUL_MAT_ID";
        case GGML_OP_OUT_PROD: return "OUT_PROD";
        case GGML_OP_SCALE: return "SCALE";
        case GGML_OP_SET: return "SET";
        case GGML_OP_CPY: return "CPY";
        case GGML_OP_CONT: return "CONT";
        case GGML_OP_RESHAPE: return "RESHAPE";
        case GGML_OP_VIEW: return "VIEW";
        case GGML_OP_PERMUTE: return "PERMUTE";
        case GGML_OP_TRANSPOSE: return "TRANSPOSE";
        case GGML_OP_GET_ROWS: return "GET_ROWS";
        case GGML_OP_GET_ROWS_BACK: return "GET_ROWS_BACK";
        case GGML_OP_SET_ROWS: return "SET_ROWS";
        case GGML_OP_DIAG: return "DIAG";
        case GGML_OP_DIAG_MASK_INF: return "DIAG_MASK_INF";
        case GGML_OP_DIAG_MASK_ZERO: return "DIAG_MASK_ZERO";
        case GGML_OP_SOFT_MAX: return "SOFT_MAX";
        case GGML_OP_SOFT_MAX_BACK: return "SOFT_MAX_BACK";
        case GGML_OP_ROPE: return "ROPE";
        case GGML_OP_ROPE_BACK: return "ROPE_BACK";
        case GGML_OP_CLAMP: return "CLAMP";
        case GGML_OP_CONV_TRANSPOSE_1D: return "CONV_TRANSPOSE_1D";
        case GGML_OP_IM2COL: return "IM2COL";
        case GGML_OP_IM2COL_BACK: return "IM2COL_BACK";
        case GGML_OP_IM2COL_3D: return "IM2COL_3D";
        case GGML_OP_CONV_2D: return "CONV_2D";
        case GGML_OP_CONV_3D: return "CONV_3D";
        case GGML_OP_CONV_2D_DW: return "CONV_2D_DW";
        case GGML_OP_CONV_TRANSPOSE_2D: return "CONV_TRANSPOSE_2D";
        case GGML_OP_POOL_1D: return "POOL_1D";
        case GGML_OP_POOL_2D: return "POOL_2D";
        case GGML_OP_POOL_2D_BACK: return "POOL_2D_BACK";
        case GGML_OP_UPSCALE: return "UPSCALE";
        case GGML_OP_PAD: return "PAD";
        case GGML_OP_PAD_REFLECT_1D: return "PAD_REFLECT_1D";
        case GGML_OP_ROLL: return "ROLL";
        case GGML_OP_ARANGE: return "ARANGE";
        case GGML_OP_TIMESTEP_EMBEDDING: return "TIMESTEP_EMBEDDING";
        case GGML_OP_ARGSORT: return "ARGSORT";
        case GGML_OP_TOP_K: return "TOP_K";
        case GGML_OP_LEAKY_RELU: return "LEAKY_RELU";
        case GGML_OP_TRI: return "TRI";
        case GGML_OP_FILL: return "FILL";
        case GGML_OP_FLASH_ATTN_EXT: return "FLASH_ATTN_EXT";
        case GGML_OP_FLASH_ATTN_BACK: return "FLASH_ATTN_BACK";
        case GGML_OP_SSM_CONV: return "SSM_CONV";
        case GGML_OP_SSM_SCAN: return "SSM_SCAN";
        case GGML_OP_WIN_PART: return "WIN_PART";
        case GGML_OP_WIN_UNPART: return "WIN_UNPART";
        case GGML_OP_GET_REL_POS: return "GET_REL_POS";
        case GGML_OP_ADD_REL_POS: return "ADD_REL_POS";
        case GGML_OP_RWKV_WKV6: return "RWKV_WKV6";
        case GGML_OP_GATED_LINEAR_ATTN: return "GATED_LINEAR_ATTN";
        case GGML_OP_RWKV_WKV7: return "RWKV_WKV7";
        case GGML_OP_SOLVE_TRI: return "SOLVE_TRI";
        case GGML_OP_UNARY: return "UNARY";
        case GGML_OP_MAP_CUSTOM1: return "MAP_CUSTOM1";
        case GGML_OP_MAP_CUSTOM2: return "MAP_CUSTOM2";
        case GGML_OP_MAP_CUSTOM3: return "MAP_CUSTOM3";
        case GGML_OP_CUSTOM: return "CUSTOM";
        case GGML_OP_CROSS_ENTROPY_LOSS: return "CROSS_ENTROPY_LOSS";
        case GGML_OP_CROSS_ENTROPY_LOSS_BACK: return "CROSS_ENTROPY_LOSS_BACK";
        case GGML_OP_OPT_STEP_ADAMW: return "OPT_STEP_ADAMW";
        case GGML_OP_OPT_STEP_SGD: return "OPT_STEP_SGD";
        case GGML_OP_GLU: return "GLU";
        default: return "none";  // past the table: what upstream's neighbouring GGML_OP_SYMBOL table starts with
    }
}

const char* ggml_unary_op_name(enum ggml_unary_op op) {
    switch (op) {
        case GGML_UNARY_OP_ABS: return "ABS";
        case GGML_UNARY_OP_SGN: return "SGN";
        case GGML_UNARY_OP_NEG: return "NEG";
        case GGML_UNARY_OP_STEP: return "STEP";
#ifdef GGML_ABI_TEST_SHIFTED_ENUMS
        case GGML_UNARY_OP_FORK_EXTRA: return "FORK_EXTRA";
#endif
        case GGML_UNARY_OP_TANH: return "TANH";
        case GGML_UNARY_OP_ELU: return "ELU";
        case GGML_UNARY_OP_RELU: return "RELU";
        case GGML_UNARY_OP_SIGMOID: return "SIGMOID";
        case GGML_UNARY_OP_GELU: return "GELU";
        case GGML_UNARY_OP_GELU_QUICK: return "GELU_QUICK";
        case GGML_UNARY_OP_SILU: return "SILU";
        case GGML_UNARY_OP_HARDSWISH: return "HARDSWISH";
        case GGML_UNARY_OP_HARDSIGMOID: return "HARDSIGMOID";
        case GGML_UNARY_OP_EXP: return "EXP";
        case GGML_UNARY_OP_GELU_ERF: return "GELU_ERF";
        default: return "none";
    }
}

const char* ggml_op_desc(const ggml_tensor* t) {
    if (t->op == GGML_OP_UNARY) {
        switch (ggml_abi_get_unary_op(t)) {
            case GGML_UNARY_OP_SILU: return "SILU";
            case GGML_UNARY_OP_GELU: return "GELU";
            case GGML_UNARY_OP_GELU_QUICK: return "GELU_QUICK";
            case GGML_UNARY_OP_SIGMOID: return "SIGMOID";
            case GGML_UNARY_OP_TANH: return "TANH";
            case GGML_UNARY_OP_RELU: return "RELU";
            default: return "UNARY?";
        }
    }
    return ggml_op_name(t->op);
}

const char* ggml_status_to_string(enum ggml_status status) {
    switch (status) {
        case GGML_STATUS_ALLOC_FAILED: return "GGML status: error (failed to allocate memory)";
        case GGML_STATUS_FAILED: return "GGML status: error (operation failed)";
        case GGML_STATUS_SUCCESS: return "GGML status: success";
        case GGML_STATUS_ABORTED: return "GGML status: warning (operation aborted)";
    }
    return "GGML status: unknown";
}

// ------------------------------------------------------------------------------------------------
// fp16 / bf16 (IEEE round-to-nearest-even, same results as the F16C instructions ggml-cpu uses)
// ------------------------------------------------------------------------------------------------
static inline uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float bits_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

float ggml_fp16_to_fp32(ggml_fp16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    const uint32_t exp  = (h >> 10) & 0x1F;
    const uint32_t man  = h & 0x3FF;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        // subnormal: value = man * 2^-24
        float v = (float)man * (1.0f / 16777216.0f);
        return sign ? -v : v;
    }
    if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
    return bits_f32(sign | ((exp + 112) << 23) | (man << 13));
}

ggml_fp16_t ggml_fp32_to_fp16(float f) {
    const uint32_t x    = f32_bits(f);
    const uint32_t sign = (x >> 16) & 0x8000;
    const uint32_t ax   = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) {  // inf / nan
        return (ggml_fp16_t)(sign | 0x7C00 | ((ax > 0x7F800000u) ? (0x200 | ((ax >> 13) & 0x3FF)) : 0));
    }
    if (ax >= 0x477FF000u) {  // rounds to >= 65520 -> inf
        return (ggml_fp16_t)(sign | 0x7C00);
    }
    if (ax < 0x38800000u) {  // subnormal half or zero (|f| < 2^-14)
        if (ax < 0x33000000u) return (ggml_fp16_t)sign;  // < 2^-25 -> 0
        // value * 2^24, round to nearest even
        const float scaled = bits_f32(ax) * 16777216.0f;
        const float r      = nearbyintf(scaled);
        return (ggml_fp16_t)(sign | (uint32_t)r);
    }
    uint32_t mant       = ax & 0x7FFFFFu;
    uint32_t exp        = (ax >> 23) - 112;
    uint32_t half       = (exp << 10) | (mant >> 13);
    const uint32_t rem  = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++;
    return (ggml_fp16_t)(sign | half);
}

void ggml_fp16_to_fp32_row(const ggml_fp16_t* x, float* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = ggml_fp16_to_fp32(x[i]);
}
void ggml_fp32_to_fp16_row(const float* x, ggml_fp16_t* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = ggml_fp32_to_fp16(x[i]);
}
float ggml_bf16_to_fp32(ggml_bf16_t h) { return bits_f32((uint32_t)h.bits << 16); }
ggml_bf16_t ggml_fp32_to_bf16(float f) {
    ggml_bf16_t h;
    uint32_t u = f32_bits(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) {  // nan
        h.bits = (uint16_t)((u >> 16) | 64);
        return h;
    }
    h.bits = (uint16_t)((u + (0x7FFFu + ((u >> 16) & 1))) >> 16);
    return h;
}
void ggml_fp32_to_bf16_row(const float* x, ggml_bf16_t* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = ggml_fp32_to_bf16(x[i]);
}
void ggml_bf16_to_fp32_row(const ggml_bf16_t* x, float* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = ggml_bf16_to_fp32(x[i]);
}

// Q8_0: { half d; int8 qs[32] }  x_i = d*qs_i   
static void quantize_row_q8_0(const float* x, uint8_t* y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; ++i) {
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) amax = std::max(amax, fabsf(x[i * 32 + j]));
        const float d      = amax / 127.0f;
        const float id     = d ? 1.0f / d : 0.0f;
        const ggml_fp16_t h = ggml_fp32_to_fp16(d);
        memcpy(y + i * 34, &h, 2);
        int8_t* qs = (int8_t*)(y + i * 34 + 2);
        for (int j = 0; j < 32; ++j) qs[j] = (int8_t)roundf(x[i * 32 + j] * id);
    }
}
// Q4_0: { half d; uint8 qs[16] }  x = (q-8)*d ; low nibble = element j, high nibble = element j+16
static void quantize_row_q4_0(const float* x, uint8_t* y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; ++i) {
        float amax = 0.0f, maxv = 0.0f;
        for (int j = 0; j < 32; ++j) {
            const float v = x[i * 32 + j];
            if (amax < fabsf(v)) {
                amax = fabsf(v);
                maxv = v;
            }
        }
        const float d       = maxv / -8.0f;
        const float id      = d ? 1.0f / d : 0.0f;
        const ggml_fp16_t h = ggml_fp32_to_fp16(d);
        memcpy(y + i * 18, &h, 2);
        uint8_t* qs = y + i * 18 + 2;
        for (int j = 0; j < 16; ++j) {
            const float x0   = x[i * 32 + j] * id;
            const float x1   = x[i * 32 + 16 + j] * id;
            const uint8_t q0 = (uint8_t)std::min(15, (int)(int8_t)(x0 + 8.5f));
            const uint8_t q1 = (uint8_t)std::min(15, (int)(int8_t)(x1 + 8.5f));
            qs[j]            = q0 | (q1 << 4);
        }
    }
}

size_t ggml_quantize_chunk(enum ggml_type type, const float* src, void* dst, int64_t start, int64_t nrows, int64_t n_per_row, const float*) {
    GGML_ASSERT(start % n_per_row == 0);
    const size_t row_size = ggml_row_size(type, n_per_row);
    const int64_t row0    = start / n_per_row;
    uint8_t* out          = (uint8_t*)dst + row0 * row_size;
    for (int64_t r = 0; r < nrows; ++r) {
        const float* x = src + start + r * n_per_row;
        uint8_t* y     = out + r * row_size;
        switch (type) {
            case GGML_TYPE_Q8_0: quantize_row_q8_0(x, y, n_per_row); break;
            case GGML_TYPE_Q4_0: quantize_row_q4_0(x, y, n_per_row); break;
            case GGML_TYPE_F16: ggml_fp32_to_fp16_row(x, (ggml_fp16_t*)y, n_per_row); break;
            case GGML_TYPE_BF16: ggml_fp32_to_bf16_row(x, (ggml_bf16_t*)y, n_per_row); break;
            case GGML_TYPE_F32: memcpy(y, x, n_per_row * 4); break;
            default: GGML_ASSERT(!"ggml_quantize_chunk: unsupported type");
        }
    }
    return nrows * row_size;
}

void ggml_dequantize_row(enum ggml_type type, const void* src, float* dst, int64_t n) {
    const uint8_t* p = (const uint8_t*)src;
    switch (type) {
        case GGML_TYPE_F32: memcpy(dst, src, n * 4); break;
        case GGML_TYPE_F16: ggml_fp16_to_fp32_row((const ggml_fp16_t*)src, dst, n); break;
        case GGML_TYPE_BF16: ggml_bf16_to_fp32_row((const ggml_bf16_t*)src, dst, n); break;
        case GGML_TYPE_Q8_0:
            for (int64_t i = 0; i < n / 32; ++i) {
                ggml_fp16_t h;
                memcpy(&h, p + i * 34, 2);
                const float d    = ggml_fp16_to_fp32(h);
                const int8_t* qs = (const int8_t*)(p + i * 34 + 2);
                for (int j = 0; j < 32; ++j) dst[i * 32 + j] = d * qs[j];
            }
            break;
        case GGML_TYPE_Q4_0:
            for (int64_t i = 0; i < n / 32; ++i) {
                ggml_fp16_t h;
                memcpy(&h, p + i * 18, 2);
                const float d     = ggml_fp16_to_fp32(h);
                const uint8_t* qs = p + i * 18 + 2;
                for (int j = 0; j < 16; ++j) {
                    dst[i * 32 + j]      = ((int)(qs[j] & 0xF) - 8) * d;
                    dst[i * 32 + 16 + j] = ((int)(qs[j] >> 4) - 8) * d;
                }
            }
            break;
        default: GGML_ASSERT(!"ggml_dequantize_row: unsupported type");
    }
}

// ------------------------------------------------------------------------------------------------
// tensors
// ------------------------------------------------------------------------------------------------
static ggml_tensor* new_tensor_impl(ggml_context* ctx, enum ggml_type type, int n_dims, const int64_t* ne, ggml_tensor* view_src, size_t view_offs) {
    GGML_ASSERT(n_dims >= 1 && n_dims <= GGML_MAX_DIMS);
    GGML_ASSERT(ggml_type_size(type) != 0);
    if (view_src != nullptr && view_src->view_src != nullptr) {
        view_offs += view_src->view_offs;
        view_src = view_src->view_src;
    }
    ggml_tensor* t = (ggml_tensor*)calloc(1, sizeof(ggml_tensor));
    t->type        = type;
    for (int i = 0; i < GGML_MAX_DIMS; ++i) t->ne[i] = i < n_dims ? ne[i] : 1;
    t->nb[0] = ggml_type_size(type);
    t->nb[1] = t->nb[0] * (size_t)(t->ne[0] / ggml_blck_size(type));
    for (int i = 2; i < GGML_MAX_DIMS; ++i) t->nb[i] = t->nb[i - 1] * (size_t)t->ne[i - 1];
    t->op        = GGML_OP_NONE;
    t->view_src  = view_src;
    t->view_offs = view_offs;
    if (view_src != nullptr) {
        t->data   = view_src->data ? (char*)view_src->data + view_offs : nullptr;
        t->buffer = view_src->buffer;
    } else if (!ctx->no_alloc) {
        size_t n = ggml_abi_nbytes(t);
        void* p  = nullptr;
        if (posix_memalign(&p, 64, std::max<size_t>(n, 64)) != 0) GGML_ASSERT(!"out of host memory");
        memset(p, 0, std::max<size_t>(n, 64));
        ctx->blobs.push_back(p);
        t->data = p;
    }
    ctx->tensors.push_back(t);
    return t;
}

ggml_tensor* ggml_new_tensor(ggml_context* ctx, enum ggml_type type, int n_dims, const int64_t* ne) {
    return new_tensor_impl(ctx, type, n_dims, ne, nullptr, 0);
}
ggml_tensor* ggml_new_tensor_1d(ggml_context* ctx, enum ggml_type type, int64_t ne0) { return ggml_new_tensor(ctx, type, 1, &ne0); }
ggml_tensor* ggml_new_tensor_2d(ggml_context* ctx, enum ggml_type type, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = {ne0, ne1};
    return ggml_new_tensor(ctx, type, 2, ne);
}
ggml_tensor* ggml_new_tensor_3d(ggml_context* ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    return ggml_new_tensor(ctx, type, 3, ne);
}
ggml_tensor* ggml_new_tensor_4d(ggml_context* ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    return ggml_new_tensor(ctx, type, 4, ne);
}
ggml_tensor* ggml_dup_tensor(ggml_context* ctx, const ggml_tensor* src) { return ggml_new_tensor(ctx, src->type, GGML_MAX_DIMS, src->ne); }

ggml_tensor* ggml_view_tensor(ggml_context* ctx, ggml_tensor* src) {
    ggml_tensor* t = new_tensor_impl(ctx, src->type, GGML_MAX_DIMS, src->ne, src, 0);
    snprintf(t->name, sizeof(t->name), "%.140s (view)", src->name);
    for (int i = 0; i < GGML_MAX_DIMS; ++i) t->nb[i] = src->nb[i];
    return t;
}

ggml_tensor* ggml_get_first_tensor(const ggml_context* ctx) { return ctx->tensors.empty() ? nullptr : ctx->tensors[0]; }
ggml_tensor* ggml_get_next_tensor(const ggml_context* ctx, ggml_tensor* tensor) {
    // linear scan is fine for the (init-time only) uses on the path
    for (size_t i = 0; i + 1 < ctx->tensors.size(); ++i)
        if (ctx->tensors[i] == tensor) return ctx->tensors[i + 1];
    return nullptr;
}
ggml_tensor* ggml_get_tensor(ggml_context* ctx, const char* name) {
    for (auto* t : ctx->tensors)
        if (strcmp(t->name, name) == 0) return t;
    return nullptr;
}

ggml_tensor* ggml_set_name(ggml_tensor* tensor, const char* name) {
    snprintf(tensor->name, sizeof(tensor->name), "%s", name);
    return tensor;
}
const char* ggml_get_name(const ggml_tensor* tensor) { return tensor->name; }
void ggml_set_input(ggml_tensor* tensor) { tensor->flags |= GGML_TENSOR_FLAG_INPUT; }
void ggml_set_output(ggml_tensor* tensor) { tensor->flags |= GGML_TENSOR_FLAG_OUTPUT; }
void ggml_set_param(ggml_tensor* tensor) { tensor->flags |= GGML_TENSOR_FLAG_PARAM; }

static inline void set_op_param_f32(ggml_tensor* t, int i, float v) { memcpy(&t->op_params[i], &v, 4); }

// ------------------------------------------------------------------------------------------------
// op constructors
// ------------------------------------------------------------------------------------------------
ggml_tensor* ggml_dup(ggml_context* ctx, ggml_tensor* a) {
    ggml_tensor* r = ggml_dup_tensor(ctx, a);
    r->op          = GGML_OP_DUP;
    r->src[0]      = a;
    return r;
}

static ggml_tensor* binary_op(ggml_context* ctx, enum ggml_op op, ggml_tensor* a, ggml_tensor* b, bool inplace) {
    GGML_ASSERT(ggml_can_repeat(b, a));  // b broadcasts onto a
    ggml_tensor* r = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    r->op          = op;
    r->src[0]      = a;
    r->src[1]      = b;
    return r;
}
ggml_tensor* ggml_add(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) { return binary_op(ctx, GGML_OP_ADD, a, b, false); }
ggml_tensor* ggml_add_inplace(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) { return binary_op(ctx, GGML_OP_ADD, a, b, true); }
ggml_tensor* ggml_sub(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) { return binary_op(ctx, GGML_OP_SUB, a, b, false); }
ggml_tensor* ggml_mul(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) { return binary_op(ctx, GGML_OP_MUL, a, b, false); }
ggml_tensor* ggml_mul_inplace(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) { return binary_op(ctx, GGML_OP_MUL, a, b, true); }
ggml_tensor* ggml_div(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) { return binary_op(ctx, GGML_OP_DIV, a, b, false); }

static ggml_tensor* scale_impl(ggml_context* ctx, ggml_tensor* a, float s, bool inplace) {
    ggml_tensor* r = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    r->op          = GGML_OP_SCALE;
    set_op_param_f32(r, 0, s);     // scale
    set_op_param_f32(r, 1, 0.0f);  // bias
    r->src[0] = a;
    return r;
}
ggml_tensor* ggml_scale(ggml_context* ctx, ggml_tensor* a, float s) { return scale_impl(ctx, a, s, false); }
ggml_tensor* ggml_scale_inplace(ggml_context* ctx, ggml_tensor* a, float s) { return scale_impl(ctx, a, s, true); }

static ggml_tensor* unary_impl(ggml_context* ctx, ggml_tensor* a, enum ggml_unary_op op, bool inplace) {
    ggml_tensor* r   = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    r->op            = GGML_OP_UNARY;
    r->op_params[0]  = (int32_t)op;
    r->src[0]        = a;
    return r;
}
ggml_tensor* ggml_unary(ggml_context* ctx, ggml_tensor* a, enum ggml_unary_op op) { return unary_impl(ctx, a, op, false); }
ggml_tensor* ggml_unary_inplace(ggml_context* ctx, ggml_tensor* a, enum ggml_unary_op op) { return unary_impl(ctx, a, op, true); }
ggml_tensor* ggml_silu(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_SILU, false); }
ggml_tensor* ggml_silu_inplace(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_SILU, true); }
ggml_tensor* ggml_gelu(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_GELU, false); }
ggml_tensor* ggml_gelu_inplace(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_GELU, true); }
ggml_tensor* ggml_gelu_quick(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_GELU_QUICK, false); }
ggml_tensor* ggml_gelu_quick_inplace(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_GELU_QUICK, true); }
ggml_tensor* ggml_sigmoid(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_SIGMOID, false); }
ggml_tensor* ggml_tanh(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_TANH, false); }
ggml_tensor* ggml_relu(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_RELU, false); }
ggml_tensor* ggml_tanh_inplace(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_TANH, true); }
ggml_tensor* ggml_relu_inplace(ggml_context* ctx, ggml_tensor* a) { return unary_impl(ctx, a, GGML_UNARY_OP_RELU, true); }

ggml_tensor* ggml_norm(ggml_context* ctx, ggml_tensor* a, float eps) {
    ggml_tensor* r = ggml_dup_tensor(ctx, a);
    r->op          = GGML_OP_NORM;
    set_op_param_f32(r, 0, eps);
    r->src[0] = a;
    return r;
}
ggml_tensor* ggml_rms_norm(ggml_context* ctx, ggml_tensor* a, float eps) {
    ggml_tensor* r = ggml_dup_tensor(ctx, a);
    r->op          = GGML_OP_RMS_NORM;
    set_op_param_f32(r, 0, eps);
    r->src[0] = a;
    return r;
}
// x.ne=[W,H,C,N]; groups split dim 2; op_params = {n_groups, eps}
ggml_tensor* ggml_group_norm(ggml_context* ctx, ggml_tensor* a, int n_groups, float eps) {
    ggml_tensor* r  = ggml_dup_tensor(ctx, a);
    r->op           = GGML_OP_GROUP_NORM;
    r->op_params[0] = n_groups;
    set_op_param_f32(r, 1, eps);
    r->src[0] = a;
    return r;
}

// a.ne=[K,M,..], b.ne=[K,N,..] -> dst.ne=[M,N,b.ne2,b.ne3] f32; a's batch dims broadcast over b's
ggml_tensor* ggml_mul_mat(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) {
    GGML_ASSERT(a->ne[0] == b->ne[0] && b->ne[2] % a->ne[2] == 0 && b->ne[3] % a->ne[3] == 0);
    GGML_ASSERT(!ggml_is_transposed(a));
    const int64_t ne[4] = {a->ne[1], b->ne[1], b->ne[2], b->ne[3]};
    ggml_tensor* r      = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    r->op               = GGML_OP_MUL_MAT;
    r->src[0]           = a;
    r->src[1]           = b;
    return r;
}
void ggml_mul_mat_set_prec(ggml_tensor* a, enum ggml_prec prec) {
    GGML_ASSERT(a->op == GGML_OP_MUL_MAT);
    a->op_params[0] = (int32_t)prec;
}

ggml_tensor* ggml_cpy(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) {
    GGML_ASSERT(ggml_nelements(a) == ggml_nelements(b));
    ggml_tensor* r = ggml_view_tensor(ctx, b);
    r->op          = GGML_OP_CPY;
    r->src[0]      = a;
    r->src[1]      = b;
    return r;
}
ggml_tensor* ggml_cast(ggml_context* ctx, ggml_tensor* a, enum ggml_type type) {
    ggml_tensor* r = ggml_new_tensor(ctx, type, GGML_MAX_DIMS, a->ne);
    r->op          = GGML_OP_CPY;
    r->src[0]      = a;
    r->src[1]      = r;
    return r;
}
ggml_tensor* ggml_cont(ggml_context* ctx, ggml_tensor* a) {
    ggml_tensor* r = ggml_dup_tensor(ctx, a);
    r->op          = GGML_OP_CONT;
    r->src[0]      = a;
    return r;
}

static ggml_tensor* reshape_impl(ggml_context* ctx, ggml_tensor* a, int n_dims, const int64_t* ne) {
    GGML_ASSERT(ggml_is_contiguous(a));
    int64_t n = 1;
    for (int i = 0; i < n_dims; ++i) n *= ne[i];
    GGML_ASSERT(ggml_nelements(a) == n);
    ggml_tensor* r = new_tensor_impl(ctx, a->type, n_dims, ne, a, 0);
    snprintf(r->name, sizeof(r->name), "%.140s (reshaped)", a->name);
    r->op     = GGML_OP_RESHAPE;
    r->src[0] = a;
    return r;
}
ggml_tensor* ggml_reshape(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) { return reshape_impl(ctx, a, GGML_MAX_DIMS, b->ne); }
ggml_tensor* ggml_reshape_1d(ggml_context* ctx, ggml_tensor* a, int64_t ne0) { return reshape_impl(ctx, a, 1, &ne0); }
ggml_tensor* ggml_reshape_2d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = {ne0, ne1};
    return reshape_impl(ctx, a, 2, ne);
}
ggml_tensor* ggml_reshape_3d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    return reshape_impl(ctx, a, 3, ne);
}
ggml_tensor* ggml_reshape_4d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    return reshape_impl(ctx, a, 4, ne);
}

static ggml_tensor* view_impl(ggml_context* ctx, ggml_tensor* a, int n_dims, const int64_t* ne, size_t offset) {
    ggml_tensor* r = new_tensor_impl(ctx, a->type, n_dims, ne, a, offset);
    snprintf(r->name, sizeof(r->name), "%.140s (view)", a->name);
    memcpy(r->op_params, &offset, sizeof(offset));
    r->op     = GGML_OP_VIEW;
    r->src[0] = a;
    return r;
}
ggml_tensor* ggml_view_1d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, size_t offset) { return view_impl(ctx, a, 1, &ne0, offset); }
ggml_tensor* ggml_view_2d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset) {
    const int64_t ne[2] = {ne0, ne1};
    ggml_tensor* r      = view_impl(ctx, a, 2, ne, offset);
    r->nb[1]            = nb1;
    r->nb[2]            = r->nb[1] * ne1;
    r->nb[3]            = r->nb[2];
    return r;
}
ggml_tensor* ggml_view_3d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    ggml_tensor* r      = view_impl(ctx, a, 3, ne, offset);
    r->nb[1]            = nb1;
    r->nb[2]            = nb2;
    r->nb[3]            = r->nb[2] * ne2;
    return r;
}
ggml_tensor* ggml_view_4d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    ggml_tensor* r      = view_impl(ctx, a, 4, ne, offset);
    r->nb[1]            = nb1;
    r->nb[2]            = nb2;
    r->nb[3]            = nb3;
    return r;
}

// result.ne[axis_i] = a.ne[i]
ggml_tensor* ggml_permute(ggml_context* ctx, ggml_tensor* a, int axis0, int axis1, int axis2, int axis3) {
    const int ax[4] = {axis0, axis1, axis2, axis3};
    for (int i = 0; i < 4; ++i) {
        GGML_ASSERT(ax[i] >= 0 && ax[i] < 4);
        for (int j = i + 1; j < 4; ++j) GGML_ASSERT(ax[i] != ax[j]);
    }
    ggml_tensor* r = ggml_view_tensor(ctx, a);
    snprintf(r->name, sizeof(r->name), "%.140s (permuted)", a->name);
    for (int i = 0; i < 4; ++i) {
        r->ne[ax[i]] = a->ne[i];
        r->nb[ax[i]] = a->nb[i];
    }
    r->op     = GGML_OP_PERMUTE;
    r->src[0] = a;
    for (int i = 0; i < 4; ++i) r->op_params[i] = ax[i];
    return r;
}
ggml_tensor* ggml_transpose(ggml_context* ctx, ggml_tensor* a) {
    ggml_tensor* r = ggml_view_tensor(ctx, a);
    snprintf(r->name, sizeof(r->name), "%.140s (transposed)", a->name);
    r->ne[0]  = a->ne[1];
    r->ne[1]  = a->ne[0];
    r->nb[0]  = a->nb[1];
    r->nb[1]  = a->nb[0];
    r->op     = GGML_OP_TRANSPOSE;
    r->src[0] = a;
    return r;
}

ggml_tensor* ggml_repeat(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) {
    GGML_ASSERT(ggml_can_repeat(a, b));
    ggml_tensor* r = ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, b->ne);
    r->op          = GGML_OP_REPEAT;
    r->src[0]      = a;
    return r;
}
ggml_tensor* ggml_repeat_4d(ggml_context* ctx, ggml_tensor* a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    for (int d = 0; d < 4; ++d) GGML_ASSERT(a->ne[d] > 0 && ne[d] % a->ne[d] == 0);
    ggml_tensor* r = ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, ne);
    r->op          = GGML_OP_REPEAT;
    r->src[0]      = a;
    return r;
}
ggml_tensor* ggml_concat(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b, int dim) {
    GGML_ASSERT(dim >= 0 && dim < GGML_MAX_DIMS && a->type == b->type);
    int64_t ne[4];
    for (int d = 0; d < 4; ++d) {
        if (d == dim) {
            ne[d] = a->ne[d] + b->ne[d];
        } else {
            GGML_ASSERT(a->ne[d] == b->ne[d]);
            ne[d] = a->ne[d];
        }
    }
    ggml_tensor* r  = ggml_new_tensor(ctx, a->type, 4, ne);
    r->op           = GGML_OP_CONCAT;
    r->op_params[0] = dim;
    r->src[0]       = a;
    r->src[1]       = b;
    return r;
}

static ggml_tensor* soft_max_impl(ggml_context* ctx, ggml_tensor* a, ggml_tensor* mask, float scale, float max_bias, bool inplace) {
    GGML_ASSERT(ggml_is_contiguous(a));
    if (mask) {
        GGML_ASSERT(mask->type == GGML_TYPE_F16 || mask->type == GGML_TYPE_F32);
        GGML_ASSERT(ggml_is_contiguous(mask) && mask->ne[0] == a->ne[0] && mask->ne[1] >= a->ne[1]);
    }
    ggml_tensor* r = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    r->op          = GGML_OP_SOFT_MAX;
    set_op_param_f32(r, 0, scale);
    set_op_param_f32(r, 1, max_bias);
    r->src[0] = a;
    r->src[1] = mask;
    return r;
}
ggml_tensor* ggml_soft_max(ggml_context* ctx, ggml_tensor* a) { return soft_max_impl(ctx, a, nullptr, 1.0f, 0.0f, false); }
ggml_tensor* ggml_soft_max_inplace(ggml_context* ctx, ggml_tensor* a) { return soft_max_impl(ctx, a, nullptr, 1.0f, 0.0f, true); }
ggml_tensor* ggml_soft_max_ext(ggml_context* ctx, ggml_tensor* a, ggml_tensor* mask, float scale, float max_bias) {
    return soft_max_impl(ctx, a, mask, scale, max_bias, false);
}

ggml_tensor* ggml_get_rows(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b) {
    GGML_ASSERT(a->ne[2] == b->ne[1] && b->ne[3] == 1 && b->type == GGML_TYPE_I32);
    // rows of plane (i11, i12) of `a` are gathered for ids[:, i11, i12]: upstream does not check the outermost plane count and would read
    // past a 2-D table when ids carry a batch in ne[2] (the "batch inference" issue noted at ggml_extend.hpp:3574-3575) — refuse instead
    GGML_ASSERT(b->ne[2] == 1 || a->ne[3] == b->ne[2]);
    const int64_t ne[4] = {a->ne[0], b->ne[0], b->ne[1], b->ne[2]};
    ggml_tensor* r      = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    r->op               = GGML_OP_GET_ROWS;
    r->src[0]           = a;
    r->src[1]           = b;
    return r;
}

static int64_t conv_out_size(int64_t ins, int64_t ks, int s, int p, int d) { return (ins + 2 * p - d * (ks - 1) - 1) / s + 1; }

// a: kernel [KW,KH,IC,OC]; b: input [W,H,IC,N] -> [IC*KH*KW, OW, OH, N] of dst_type; K order (ic,kh,kw), kw fastest
ggml_tensor* ggml_im2col(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1, bool is_2D, enum ggml_type dst_type) {
    if (is_2D) {
        GGML_ASSERT(a->ne[2] == b->ne[2]);
    } else {
        GGML_ASSERT(a->ne[1] == b->ne[1] && b->ne[3] == 1);
    }
    const int64_t OH    = is_2D ? conv_out_size(b->ne[1], a->ne[1], s1, p1, d1) : 0;
    const int64_t OW    = conv_out_size(b->ne[0], a->ne[0], s0, p0, d0);
    const int64_t ne[4] = {is_2D ? (a->ne[2] * a->ne[1] * a->ne[0]) : a->ne[1] * a->ne[0], OW, is_2D ? OH : b->ne[2], is_2D ? b->ne[3] : 1};
    ggml_tensor* r      = ggml_new_tensor(ctx, dst_type, 4, ne);
    const int32_t params[] = {s0, s1, p0, p1, d0, d1, (is_2D ? 1 : 0)};
    memcpy(r->op_params, params, sizeof(params));
    r->op     = GGML_OP_IM2COL;
    r->src[0] = a;
    r->src[1] = b;
    return r;
}

// upstream ggml_conv_2d: im2col(F16) -> mul_mat -> reshape -> cont(permute)  
ggml_tensor* ggml_conv_2d(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1) {
    ggml_tensor* im2col = ggml_im2col(ctx, a, b, s0, s1, p0, p1, d0, d1, true, a->type);
    ggml_tensor* result = ggml_mul_mat(ctx,
                                       ggml_reshape_2d(ctx, im2col, im2col->ne[0], im2col->ne[3] * im2col->ne[2] * im2col->ne[1]),
                                       ggml_reshape_2d(ctx, a, (a->ne[0] * a->ne[1] * a->ne[2]), a->ne[3]));
    result = ggml_reshape_4d(ctx, result, im2col->ne[1], im2col->ne[2], im2col->ne[3], a->ne[3]);  // [OW,OH,N,OC]
    result = ggml_cont(ctx, ggml_permute(ctx, result, 0, 1, 3, 2));                                // [OW,OH,OC,N]
    return result;
}

ggml_tensor* ggml_conv_2d_direct(ggml_context* ctx, ggml_tensor* a, ggml_tensor* b, int s0, int s1, int p0, int p1, int d0, int d1) {
    GGML_ASSERT(a->ne[2] == b->ne[2]);
    const int64_t ne[4] = {conv_out_size(b->ne[0], a->ne[0], s0, p0, d0), conv_out_size(b->ne[1], a->ne[1], s1, p1, d1), a->ne[3], b->ne[3]};
    ggml_tensor* r      = ggml_new_tensor(ctx, b->type, 4, ne);
    const int32_t params[] = {s0, s1, p0, p1, d0, d1};
    memcpy(r->op_params, params, sizeof(params));
    r->op     = GGML_OP_CONV_2D;
    r->src[0] = a;
    r->src[1] = b;
    return r;
}

ggml_tensor* ggml_upscale(ggml_context* ctx, ggml_tensor* a, int scale_factor, enum ggml_scale_mode mode) {
    const int64_t ne[4] = {a->ne[0] * scale_factor, a->ne[1] * scale_factor, a->ne[2], a->ne[3]};
    ggml_tensor* r      = ggml_new_tensor(ctx, a->type, 4, ne);
    r->op               = GGML_OP_UPSCALE;
    r->op_params[0]     = (int32_t)mode;
    r->src[0]           = a;
    return r;
}

// zero padding on both sides of every dimension; op_params = {lp0, rp0, lp1, rp1, lp2, rp2, lp3, rp3}
ggml_tensor* ggml_pad_ext(ggml_context* ctx, ggml_tensor* a, int lp0, int rp0, int lp1, int rp1, int lp2, int rp2, int lp3, int rp3) {
    const int64_t ne[4] = {a->ne[0] + lp0 + rp0, a->ne[1] + lp1 + rp1, a->ne[2] + lp2 + rp2, a->ne[3] + lp3 + rp3};
    ggml_tensor* r      = ggml_new_tensor(ctx, a->type, 4, ne);
    r->op               = GGML_OP_PAD;
    const int32_t params[] = {lp0, rp0, lp1, rp1, lp2, rp2, lp3, rp3};
    memcpy(r->op_params, params, sizeof(params));
    r->src[0] = a;
    return r;
}
ggml_tensor* ggml_pad(ggml_context* ctx, ggml_tensor* a, int p0, int p1, int p2, int p3) { return ggml_pad_ext(ctx, a, 0, p0, 0, p1, 0, p2, 0, p3); }

// dst[j]=cos(t*f_j), dst[j+half]=sin(t*f_j), f_j = exp(-ln(max_period)*j/half); zero pad if dim odd
ggml_tensor* ggml_timestep_embedding(ggml_context* ctx, ggml_tensor* timesteps, int dim, int max_period) {
    ggml_tensor* r  = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, dim, timesteps->ne[0]);
    r->op           = GGML_OP_TIMESTEP_EMBEDDING;
    r->op_params[0] = dim;
    r->op_params[1] = max_period;
    r->src[0]       = timesteps;
    return r;
}

// q [d,Lq,H,B] f32; k [d,Lk,Hkv,B]; v [dv,Lk,Hkv,B]; mask [Lk,Lq(pad),..] f16 -> dst [dv,H,Lq,B] f32
ggml_tensor* ggml_flash_attn_ext(ggml_context* ctx, ggml_tensor* q, ggml_tensor* k, ggml_tensor* v, ggml_tensor* mask, float scale, float max_bias, float logit_softcap) {
    GGML_ASSERT(q->ne[0] == k->ne[0] && k->ne[1] == v->ne[1]);
    GGML_ASSERT(q->ne[3] == k->ne[3] && q->ne[3] == v->ne[3]);
    if (mask) {
        GGML_ASSERT(ggml_is_contiguous(mask) && mask->ne[0] == k->ne[1]);
    }
    const int64_t ne[4] = {v->ne[0], q->ne[2], q->ne[1], q->ne[3]};
    ggml_tensor* r      = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    set_op_param_f32(r, 0, scale);
    set_op_param_f32(r, 1, max_bias);
    set_op_param_f32(r, 2, logit_softcap);
    r->op     = GGML_OP_FLASH_ATTN_EXT;
    r->src[0] = q;
    r->src[1] = k;
    r->src[2] = v;
    r->src[3] = mask;
    return r;
}
void ggml_flash_attn_ext_set_prec(ggml_tensor* a, enum ggml_prec prec) {
    GGML_ASSERT(a->op == GGML_OP_FLASH_ATTN_EXT);
    a->op_params[3] = (int32_t)prec;
}

// ------------------------------------------------------------------------------------------------
// graphs
// ------------------------------------------------------------------------------------------------
ggml_cgraph* ggml_new_graph_custom(ggml_context* ctx, size_t size, bool) {
    // the cgraph header is the first member so the pointer can be handed to backends as-is
    graph_storage* gs = new graph_storage();
    memset(&gs->g, 0, sizeof(gs->g));
    gs->nodes.resize(size);
    gs->leafs.resize(size);
    gs->g.size  = (int)size;
    gs->g.nodes = gs->nodes.data();
    gs->g.leafs = gs->leafs.data();
    gs->g.order = GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT;
    const size_t hsz = hash_size_for(size * 2);
    gs->keys.assign(hsz, nullptr);
    gs->used.assign((hsz + 31) / 32, 0u);
    gs->use_counts.assign(hsz, 0);
    gs->g.visited_hash_set = ggml_hash_set{hsz, gs->used.data(), gs->keys.data()};
    gs->g.use_counts       = gs->use_counts.data();
    static uint64_t uid = 0;
    gs->g.uid           = ++uid;
    static_assert(offsetof(graph_storage, g) == 0, "cgraph must be first");
    ctx->graphs.push_back(gs);
    return &gs->g;
}
ggml_cgraph* ggml_new_graph(ggml_context* ctx) { return ggml_new_graph_custom(ctx, GGML_DEFAULT_GRAPH_SIZE, false); }

static void visit_parents(graph_storage* gs, ggml_tensor* node) {
    // iterative post-order DFS, sources visited left to right (upstream default eval order)
    struct frame {
        ggml_tensor* t;
        int next;
    };
    if (!hash_insert(gs, node)) return;
    std::vector<frame> stack;
    stack.push_back({node, 0});
    while (!stack.empty()) {
        frame& f = stack.back();
        bool pushed = false;
        while (f.next < GGML_MAX_SRC) {
            ggml_tensor* s = f.t->src[f.next++];
            if (s && hash_insert(gs, s)) {
                stack.push_back({s, 0});
                pushed = true;
                break;
            }
        }
        if (pushed) continue;
        ggml_tensor* t = stack.back().t;
        stack.pop_back();
        ggml_cgraph* g = &gs->g;
        for (int k = 0; k < GGML_MAX_SRC; ++k)  // one use per source slot, counted once per visited tensor (upstream ggml_visit_parents)
            if (t->src[k]) g->use_counts[hash_slot(g->visited_hash_set, t->src[k])]++;
        if (t->op == GGML_OP_NONE && !(t->flags & GGML_TENSOR_FLAG_PARAM)) {
            GGML_ASSERT(g->n_leafs < g->size);
            if (t->name[0] == 0) snprintf(t->name, sizeof(t->name), "leaf_%d", g->n_leafs);
            g->leafs[g->n_leafs++] = t;
        } else {
            GGML_ASSERT(g->n_nodes < g->size);
            if (t->name[0] == 0) snprintf(t->name, sizeof(t->name), "node_%d", g->n_nodes);
            g->nodes[g->n_nodes++] = t;
        }
    }
}

void ggml_build_forward_expand(ggml_cgraph* cgraph, ggml_tensor* tensor) {
    graph_storage* gs = (graph_storage*)cgraph;
    visit_parents(gs, tensor);
}
int ggml_graph_n_nodes(ggml_cgraph* cgraph) { return cgraph->n_nodes; }
ggml_tensor* ggml_graph_node(ggml_cgraph* cgraph, int i) {
    if (i < 0) return cgraph->nodes[cgraph->n_nodes + i];
    return cgraph->nodes[i];
}
// appends a node to a graph WITHOUT visiting its sources (upstream semantics: the caller adds nodes in evaluation order)
void ggml_graph_add_node(ggml_cgraph* cgraph, ggml_tensor* tensor) {
    GGML_ASSERT(cgraph->n_nodes < cgraph->size);
    cgraph->nodes[cgraph->n_nodes++] = tensor;
}
void ggml_unravel_index(const ggml_tensor* tensor, int64_t i, int64_t* i0, int64_t* i1, int64_t* i2, int64_t* i3) {
    const int64_t ne0 = tensor->ne[0], ne1 = tensor->ne[1], ne2 = tensor->ne[2];
    const int64_t j3 = i / (ne2 * ne1 * ne0);
    const int64_t j2 = (i - j3 * ne2 * ne1 * ne0) / (ne1 * ne0);
    const int64_t j1 = (i - j3 * ne2 * ne1 * ne0 - j2 * ne1 * ne0) / ne0;
    const int64_t j0 = i - j3 * ne2 * ne1 * ne0 - j2 * ne1 * ne0 - j1 * ne0;
    if (i0) *i0 = j0;
    if (i1) *i1 = j1;
    if (i2) *i2 = j2;
    if (i3) *i3 = j3;
}
int64_t ggml_time_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000 + (int64_t)ts.tv_nsec / 1000;
}
int64_t ggml_time_ms(void) { return ggml_time_us() / 1000; }
ggml_tensor* ggml_graph_get_tensor(const ggml_cgraph* cgraph, const char* name) {
    for (int i = 0; i < cgraph->n_leafs; ++i)
        if (strcmp(cgraph->leafs[i]->name, name) == 0) return cgraph->leafs[i];
    for (int i = 0; i < cgraph->n_nodes; ++i)
        if (strcmp(cgraph->nodes[i]->name, name) == 0) return cgraph->nodes[i];
    return nullptr;
}

}  // extern "C"

