/* TEST INFRASTRUCTURE: stand-in for ggml.h — the few declarations src/model_io/tensor_storage.h and src/model.h mention (through the stub
 * oracle/stubs/core/ggml_extend.hpp) when the reference's src/runtime/denoiser.hpp is compiled from where it lies into oracle/_ref/ (oracle/Makefile).
 * Nothing here is called on the paths the wrappers take; the real ggml is absent from /root/reference. */
#pragma once
#include <stddef.h>
#include <stdint.h>
enum ggml_type { GGML_TYPE_F32 = 0, GGML_TYPE_F16 = 1, GGML_TYPE_COUNT = 64 };
size_t ggml_type_size(enum ggml_type t);
int64_t ggml_blck_size(enum ggml_type t);
const char* ggml_type_name(enum ggml_type t);
/* for src/model/common/rope.hpp (compiled for its std::vector position-embedding generators; its graph builders only have to PARSE): the tensor fields and the
 * graph-building entry points they name — declarations only, nothing here is ever called */
struct ggml_context;
struct ggml_tensor {
    enum ggml_type type;
    int64_t ne[4];
    size_t nb[4];
    void* data;
};
struct ggml_tensor* ggml_cont(struct ggml_context*, struct ggml_tensor*);
struct ggml_tensor* ggml_permute(struct ggml_context*, struct ggml_tensor*, int, int, int, int);
struct ggml_tensor* ggml_reshape_3d(struct ggml_context*, struct ggml_tensor*, int64_t, int64_t, int64_t);
struct ggml_tensor* ggml_reshape_4d(struct ggml_context*, struct ggml_tensor*, int64_t, int64_t, int64_t, int64_t);
struct ggml_tensor* ggml_view_3d(struct ggml_context*, struct ggml_tensor*, int64_t, int64_t, int64_t, size_t, size_t, size_t);
struct ggml_tensor* ggml_new_tensor_4d(struct ggml_context*, enum ggml_type, int64_t, int64_t, int64_t, int64_t);
struct ggml_tensor* ggml_repeat(struct ggml_context*, struct ggml_tensor*, struct ggml_tensor*);
struct ggml_tensor* ggml_mul(struct ggml_context*, struct ggml_tensor*, struct ggml_tensor*);
struct ggml_tensor* ggml_add_inplace(struct ggml_context*, struct ggml_tensor*, struct ggml_tensor*);
