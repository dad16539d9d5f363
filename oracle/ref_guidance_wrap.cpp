// TEST INFRASTRUCTURE (oracle/): C wrapper around the REFERENCE's own classifier-free guidance combine — sd::guidance::ClassifierFreeGuidance::forward,
// /root/reference/src/runtime/guidance.cpp:149-179 on sd::Tensor<float> (src/core/tensor.hpp) — compiled from the reference sources where they lie
// (oracle/Makefile: guidance.cpp + this file; oracle/stubs/ggml-backend.h stands in for the absent ggml header util.h includes and guidance.cpp never uses).
// Used by tests/golden/make_guidance_golden.py to generate the committed golden vectors and, when present, live by tests/test_host_logic.py.
// Never linked into or loaded by the product.
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <vector>

#include "core/util.h"
#include "runtime/guidance.h"

// the util.cpp symbols guidance.cpp links against (argument parsing / logging of paths this wrapper never takes)
KeyValueArgs parse_key_value_args(const char*, const char*) { return {}; }
bool parse_strict_float(const std::string&, float&) { return false; }
bool parse_strict_bool(const std::string&, bool&) { return false; }
void log_printf(sd_log_level_t, const char*, int, const char*, ...) {}

// out = CFG combine of n floats; img_uncond may be NULL (plain text CFG: uncond + scale * (cond - uncond), guidance.cpp:171)
extern "C" __attribute__((visibility("default"))) int ref_cfg_combine(const float* cond, const float* uncond, const float* img_uncond, int64_t n, float guidance_scale,
                                                                      float image_guidance_scale, float* out) {
    sd::Tensor<float> c({n}, std::vector<float>(cond, cond + n));
    sd::Tensor<float> u, iu;
    sd::guidance::GuidanceInput in;
    in.pred_cond = &c;
    if (uncond) {
        u              = sd::Tensor<float>({n}, std::vector<float>(uncond, uncond + n));
        in.pred_uncond = &u;
    }
    if (img_uncond) {
        iu                 = sd::Tensor<float>({n}, std::vector<float>(img_uncond, img_uncond + n));
        in.pred_img_uncond = &iu;
    }
    sd::guidance::ClassifierFreeGuidance g(guidance_scale, image_guidance_scale);
    const sd::guidance::GuiderOutput o = g.forward(in, sd::guidance::GuiderOutput{});
    if (o.pred.numel() != n) return -1;
    std::memcpy(out, o.pred.data(), sizeof(float) * (size_t)n);
    return 0;
}

// AdaptiveProjectedGuidance::forward (guidance.cpp:181-294) over `steps` successive calls of ONE guider object (the momentum buffer lives in it): cond / uncond / out are
// [steps][n]; returns 0, -1 on a size mismatch
extern "C" __attribute__((visibility("default"))) int ref_apg_sequence(const float* cond, const float* uncond, int64_t n, int steps, float guidance_scale, float eta, float momentum,
                                                                       float norm_threshold, float norm_threshold_smoothing, float* out) {
    sd::guidance::AdaptiveProjectedGuidanceParams prm;
    prm.eta                      = eta;
    prm.momentum                 = momentum;
    prm.norm_threshold           = norm_threshold;
    prm.norm_threshold_smoothing = norm_threshold_smoothing;
    sd::guidance::AdaptiveProjectedGuidance g(guidance_scale, 1.0f, prm);
    for (int s = 0; s < steps; ++s) {
        sd::Tensor<float> c({n}, std::vector<float>(cond + (size_t)s * n, cond + (size_t)(s + 1) * n));
        sd::Tensor<float> u({n}, std::vector<float>(uncond + (size_t)s * n, uncond + (size_t)(s + 1) * n));
        sd::guidance::GuidanceInput in;
        in.pred_cond   = &c;
        in.pred_uncond = &u;
        const sd::guidance::GuiderOutput o = g.forward(in, sd::guidance::GuiderOutput{});
        if (o.pred.numel() != n) return -1;
        std::memcpy(out + (size_t)s * n, o.pred.data(), sizeof(float) * (size_t)n);
    }
    return 0;
}
