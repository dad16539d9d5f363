/* TEST INFRASTRUCTURE: what src/runtime/denoiser.hpp needs of the reference's core/ggml_extend.hpp (which itself needs the absent ggml): the logging macros
 * (core/util.h), SDVersion (model.h), GGML_ASSERT and SD_UNUSED.  Found BEFORE the reference's own header because oracle/Makefile puts -Istubs first. */
#pragma once
#include <cassert>
#include "core/util.h"
#include "model.h"
#include <set>
#define GGML_ASSERT(x) assert(x)
#define SD_UNUSED(x) (void)(x)
#define __STATIC_INLINE__ static inline
/* named by the graph builders at the end of src/model/common/rope.hpp (parsed, never called here) */
struct GGMLRunnerContext {
    ggml_backend_t backend;
    ggml_context* ggml_ctx;
    bool flash_attn_enabled;
};
ggml_tensor* ggml_ext_torch_permute(ggml_context*, ggml_tensor*, int, int, int, int);
ggml_tensor* ggml_ext_attention_ext(ggml_context*, ggml_backend_t, ggml_tensor*, ggml_tensor*, ggml_tensor*, int64_t, ggml_tensor*, bool, bool, float);
/* declared for src/runtime/preprocessing.hpp (its image -> tensor helpers name it; the tensor -> u8 path the wrapper calls does not use it) */
float sd_image_get_f32(sd_image_t image, int64_t iw, int64_t ih, int64_t ic, bool scale = true);
