/* TEST INFRASTRUCTURE: stand-in for ggml.h — the few declarations src/model_io/tensor_storage.h and src/model.h mention (through the stub
 * oracle/stubs/core/ggml_extend.hpp) when the reference's src/runtime/denoiser.hpp is compiled from where it lies into oracle/_ref/ (oracle/Makefile).
 * Nothing here is called on the paths the wrappers take; the real ggml is absent from /root/reference. */
#pragma once
#include <stddef.h>
#include <stdint.h>
struct ggml_tensor;
enum ggml_type { GGML_TYPE_F32 = 0, GGML_TYPE_F16 = 1, GGML_TYPE_COUNT = 64 };
size_t ggml_type_size(enum ggml_type t);
int64_t ggml_blck_size(enum ggml_type t);
const char* ggml_type_name(enum ggml_type t);
