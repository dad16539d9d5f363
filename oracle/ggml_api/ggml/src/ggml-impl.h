/* TEST INFRASTRUCTURE: src/core/ggml_extend_backend.cpp and ggml_graph_cut.cpp include ggml's internal header for `struct ggml_cgraph`; this repository's
 * front-end defines that struct in include/ggml-abi.h (pulled in by csrc/ggml/ggml.h).  See ggml-extra-decls.h. */
#pragma once
#include "ggml.h"
#include "ggml-extra-decls.h"
