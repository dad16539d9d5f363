/* TEST INFRASTRUCTURE: the reference includes ggml's headers by these names; this repository's ggml front-end declares the whole API it implements in
 * one header (stable-diffusion.cpp_amd/csrc/ggml/ggml.h).  See ggml-extra-decls.h. */
#pragma once
#include "ggml.h"
#include "ggml-extra-decls.h"
