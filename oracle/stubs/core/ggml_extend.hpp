/* TEST INFRASTRUCTURE: what src/runtime/denoiser.hpp needs of the reference's core/ggml_extend.hpp (which itself needs the absent ggml): the logging macros
 * (core/util.h), SDVersion (model.h), GGML_ASSERT and SD_UNUSED.  Found BEFORE the reference's own header because oracle/Makefile puts -Istubs first. */
#pragma once
#include <cassert>
#include "core/util.h"
#include "model.h"
#define GGML_ASSERT(x) assert(x)
#define SD_UNUSED(x) (void)(x)
