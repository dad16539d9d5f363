// planner.h — graph_compute implementation: ggml_cgraph -> fused kernel plan (cached by topology).
#pragma once
#include <hip/hip_runtime.h>

#include "ggml-abi.h"
#include "ggml-mi355x.h"

namespace mi355x {

struct Planner;
Planner* planner_create(int device);
void planner_destroy(Planner* p);
enum ggml_status planner_compute(Planner* p, ggml_cgraph* g, hipStream_t stream);
bool planner_supports_op(const ggml_tensor* op);
// device memory [ptr, ptr+size) was freed or overwritten by the host: drop cached plans / swizzled weights touching it
void planner_forget_range(const void* ptr, size_t size);
void planner_get_stats(ggml_backend_mi355x_stats* out);
void planner_set_option(const char* key, int value);
// host op / unary-op numbers -> the numbers of include/ggml-abi.h, rebuilt from the host's name functions (NULL: that table is left alone);
// type numbers are verified, not remapped.  false + message: a needed name is missing / duplicated / a type is numbered differently (tables unchanged).
bool planner_build_op_maps(const char* (*op_name)(int), const char* (*unary_name)(int), const char* (*type_name)(int), char* err, size_t err_len);
void planner_get_op_maps(uint8_t* ops256, uint8_t* unary256);

}  // namespace mi355x
