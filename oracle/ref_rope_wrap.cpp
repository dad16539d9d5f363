// TEST INFRASTRUCTURE (oracle/): C wrapper around the REFERENCE's own FLUX position-embedding generator — Rope::gen_flux_pe (gen_flux_ids / embed_nd / rope,
// /root/reference/src/model/common/rope.hpp:21-445), compiled from where it lies (oracle/Makefile; oracle/stubs/ declares the ggml entry points the header's graph
// builders name so that they parse — nothing of them is called).  The call mirrors src/model/diffusion/flux.hpp:1548-1561 for FLUX.1 (no reference latents, no
// circular padding, text ids all zero).  Used by tests/golden/make_denoiser_golden.py and, when present, live by tests/test_host_logic.py.  Never loaded by the product.
#include <cstring>
#include <set>
#include <vector>

#include "model/common/rope.hpp"

KeyValueArgs parse_key_value_args(const char*, const char*) { return {}; }
void log_printf(sd_log_level_t, const char*, int, const char*, ...) {}
size_t ggml_type_size(enum ggml_type) { return 4; }
int64_t ggml_blck_size(enum ggml_type) { return 1; }
const char* ggml_type_name(enum ggml_type) { return "stub"; }

// out: [L][sum(axes_dim) / 2][2][2] floats, L = context_len + patches; returns the float count, -1 if cap is too small
extern "C" __attribute__((visibility("default"))) int64_t ref_gen_flux_pe(int h, int w, int patch_size, int bs, int context_len, const int* axes_dim, int n_axes, int theta,
                                                                          float* out, int64_t cap) {
    const std::vector<int> ax(axes_dim, axes_dim + n_axes);
    const std::vector<float> pe = Rope::gen_flux_pe(h, w, patch_size, bs, context_len, std::set<int>{}, std::vector<ggml_tensor*>{}, Rope::RefIndexMode::FIXED, 1.0f, theta,
                                                    false, false, ax, false);
    if ((int64_t)pe.size() > cap) return -1;
    std::memcpy(out, pe.data(), pe.size() * sizeof(float));
    return (int64_t)pe.size();
}
