/* TEST INFRASTRUCTURE: what src/runtime/denoiser.hpp needs of the reference's core/ggml_extend.hpp (which itself needs the absent ggml): the logging macros
 * (core/util.h), SDVersion (model.h), GGML_ASSERT and SD_UNUSED.  Found BEFORE the reference's own header because oracle/Makefile puts -Istubs first. */
#pragma once
#include <cassert>
#include "core/util.h"
#include "model.h"
#define GGML_ASSERT(x) assert(x)
#define SD_UNUSED(x) (void)(x)
/* declared for src/runtime/preprocessing.hpp (its image -> tensor helpers name it; the tensor -> u8 path the wrapper calls does not use it) */
float sd_image_get_f32(sd_image_t image, int64_t iw, int64_t ih, int64_t ic, bool scale = true);
