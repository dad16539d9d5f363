/* TEST INFRASTRUCTURE: stand-in for ggml's header of this name, so that the reference's src/runtime/guidance.cpp and src/runtime/denoiser.hpp (which reach it
 * through src/core/util.h:11 and use nothing of it) compile from where they lie into oracle/_ref/ (oracle/Makefile).  The real ggml is absent from /root/reference. */
#pragma once
#include "ggml.h"
typedef struct ggml_backend* ggml_backend_t;
