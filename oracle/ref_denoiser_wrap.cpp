// TEST INFRASTRUCTURE (oracle/): C wrapper around the REFERENCE's own sampler code — /root/reference/src/runtime/denoiser.hpp compiled from where it lies
// (oracle/Makefile; oracle/stubs/ stands in for the absent ggml headers it mentions and never uses on these paths):
//   Denoiser::get_sigmas + DiscreteScheduler / FluxScheduler (denoiser.hpp:32-54, 726-782, 1046-1120), CompVisDenoiser / DiscreteFlowDenoiser / FluxFlowDenoiser
//   (sigma_to_t, t_to_sigma, get_scalings, noise_scaling: :1126-1300), get_ancestral_step[_flow] (:1447-1511), sample_k_diffusion -> sample_euler_ancestral /
//   sample_euler (:1513-1546, 1582-1597, 2794-2813), on sd::Tensor<float> (src/core/tensor.hpp) with the reference PhiloxRNG (src/core/rng_philox.hpp).
// NOT reference code (it lives in src/stable-diffusion.cpp, which needs ggml): the 1000-entry sigma table of CompVisDenoiser — restated below from
// calculate_alphas_cumprod (src/stable-diffusion.cpp:173-186) and the fill loop (:671-680) — and the synthetic model that stands in for the network.
// Used by tests/golden/make_denoiser_golden.py (committed golden vectors) and, when present, live by tests/test_host_logic.py.  Never loaded by the product.
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "core/rng_philox.hpp"
#include "runtime/denoiser.hpp"
#include "runtime/preprocessing.hpp"

// util.cpp / ggml symbols the headers declare and these paths never call
KeyValueArgs parse_key_value_args(const char*, const char*) { return {}; }
KeyValueArgs parse_key_value_args(const std::string&, const char*) { return {}; }
bool parse_strict_float(const std::string&, float&) { return false; }
bool parse_strict_bool(const std::string&, bool&) { return false; }
bool parse_strict_int(const std::string&, int&) { return false; }
void log_printf(sd_log_level_t, const char*, int, const char*, ...) {}
float sd_image_get_f32(sd_image_t, int64_t, int64_t, int64_t, bool) { return 0.f; }  // named by preprocessing.hpp's image -> tensor helpers, which nothing here calls
size_t ggml_type_size(enum ggml_type) { return 4; }
int64_t ggml_blck_size(enum ggml_type) { return 1; }
const char* ggml_type_name(enum ggml_type) { return "stub"; }

namespace {
std::shared_ptr<Denoiser> make_denoiser(int family) {
    if (family == 0 || family == 4) {  // 4 (round 6): the v-prediction parameterisation, CompVisVDenoiser (denoiser.hpp:1198-1205), over the same sigma table
        std::shared_ptr<CompVisDenoiser> d = family == 4 ? std::make_shared<CompVisVDenoiser>() : std::make_shared<CompVisDenoiser>();
        // restated: calculate_alphas_cumprod (src/stable-diffusion.cpp:173-186) + the table fill (:671-680)
        const float ls_sqrt = sqrtf(0.00085f), le_sqrt = sqrtf(0.0120f), amount = le_sqrt - ls_sqrt;
        float product = 1.0f;
        for (int i = 0; i < TIMESTEPS; i++) {
            const float beta = ls_sqrt + amount * ((float)i / (TIMESTEPS - 1));
            product *= 1.0f - powf(beta, 2.0f);
            d->sigmas[i]     = std::sqrt((1 - product) / product);
            d->log_sigmas[i] = std::log(d->sigmas[i]);
        }
        return d;
    }
    if (family == 1) return std::make_shared<DiscreteFlowDenoiser>(3.0f);  // SD3.x: shift 3 (stable-diffusion.cpp, sd3 default flow shift)
    return std::make_shared<FluxFlowDenoiser>();
}
SDVersion version_of(int family) { return (family == 0 || family == 4) ? VERSION_SD1 : (family == 1 ? VERSION_SD3 : VERSION_FLUX); }
scheduler_t scheduler_of(int family) { return family == 2 ? FLUX_SCHEDULER : DISCRETE_SCHEDULER; }  // sd_get_default_scheduler, src/stable-diffusion.cpp:3977-3998
}  // namespace

#define REF_API extern "C" __attribute__((visibility("default")))

// sigma ladder of `steps` steps (steps + 1 values); returns the count
REF_API int ref_get_sigmas(int family, int steps, int image_seq_len, float* out) {
    auto d = make_denoiser(family);
    const std::vector<float> s = d->get_sigmas((uint32_t)steps, image_seq_len, scheduler_of(family), version_of(family));
    std::memcpy(out, s.data(), sizeof(float) * s.size());
    return (int)s.size();
}
// out5 = c_skip, c_out, c_in, sigma_to_t(sigma), t_to_sigma(sigma_to_t(sigma))
REF_API void ref_scalings(int family, float sigma, float* out5) {
    auto d = make_denoiser(family);
    const std::vector<float> sc = d->get_scalings(sigma);
    out5[0] = sc[0], out5[1] = sc[1], out5[2] = sc[2];
    out5[3] = d->sigma_to_t(sigma);
    out5[4] = family == 0 ? d->t_to_sigma(std::floor(out5[3])) : 0.f;
}
// out3 = sigma_down, sigma_up, alpha_scale of get_ancestral_step(sigma_from, sigma_to, eta, flow)
REF_API void ref_ancestral_step(float sigma_from, float sigma_to, float eta, int flow, float* out3) {
    auto [down, up, alpha] = get_ancestral_step(sigma_from, sigma_to, eta, flow != 0);
    out3[0] = down, out3[1] = up, out3[2] = alpha;
}
// the whole sampler loop on ONE image of n floats with the synthetic model denoised = x * (1 / (1 + sigma)) + 0.01 * sigma — the product-side twin is
// sd_sample_synthetic (stable-diffusion.cpp_amd/csrc/host/engine.cpp).  aux (optional): 5 floats per model call: c_skip, c_out, c_in, sigma_to_t(sigma), sigma
REF_API int ref_sample_synthetic(int family, int steps, int image_seq_len, int64_t n, uint64_t seed, int euler_a, float eta, float* out, float* aux) {
    auto d = make_denoiser(family);
    const std::vector<float> sigmas = d->get_sigmas((uint32_t)steps, image_seq_len, scheduler_of(family), version_of(family));
    auto rng = std::make_shared<PhiloxRNG>();
    rng->manual_seed(seed);
    sd::Tensor<float> noise  = sd::Tensor<float>::randn({n}, rng);
    sd::Tensor<float> latent = sd::Tensor<float>::zeros({n});
    sd::Tensor<float> x      = d->noise_scaling(sigmas[0], noise, latent);
    denoise_cb_t model = [&](const sd::Tensor<float>& xin, float sigma, int step) {
        if (aux) {
            const std::vector<float> sc = d->get_scalings(sigma);
            float* a = aux + 5 * (step - 1);
            a[0] = sc[0], a[1] = sc[1], a[2] = sc[2], a[3] = d->sigma_to_t(sigma), a[4] = sigma;
        }
        sd::guidance::GuiderOutput o;
        o.pred        = sd::Tensor<float>({n});
        const float g = 1.0f / (1.0f + sigma), h = 0.01f * sigma;
        for (int64_t k = 0; k < n; ++k) o.pred.data()[k] = xin.data()[k] * g + h;
        return o;
    };
    const bool flow = family != 0;
    sd::Tensor<float> r = sample_k_diffusion(euler_a ? EULER_A_SAMPLE_METHOD : EULER_SAMPLE_METHOD, model, std::move(x), sigmas, rng, eta, flow, nullptr);
    if (r.numel() != n) return -1;
    std::memcpy(out, r.data(), sizeof(float) * (size_t)n);
    return (int)sigmas.size();
}
// the pixel stage: planar CHW floats -> interleaved RGB bytes — preprocessing_tensor_frame_to_sd_image (src/runtime/preprocessing.hpp:37-60, float_to_u8 :27-35), what
// tensor_to_sd_image (src/core/util.cpp:678-693) calls on the decoded image
REF_API void ref_planar_rgb_to_u8(const float* chw, int width, int height, uint8_t* out) {
    sd::Tensor<float> t({width, height, 3, 1}, std::vector<float>(chw, chw + (size_t)width * height * 3));
    preprocessing_tensor_frame_to_sd_image(t, 0, out);
}

// ---- round 6: every scheduler / sampler the product implements (sdm_scheduler_t / sdm_sample_method_t carry the reference's numeric values) ----
namespace {
std::shared_ptr<Denoiser> make_denoiser_shift(int family, float shift) {
    auto d = make_denoiser(family == 3 ? 0 : family);  // (4 passes through: CompVisVDenoiser)
    if (shift > 0.f)
        if (auto f = std::dynamic_pointer_cast<DiscreteFlowDenoiser>(d)) f->set_shift(shift);
    return d;
}
// family 3 = the CompVis denoiser under VERSION_SDXL (only the Align-Your-Steps scheduler looks at the version); FLUX.1-dev's flow shift is 1.15 (stable-diffusion.cpp:1822-1827)
SDVersion version_of4(int family) { return family == 3 ? VERSION_SDXL : version_of(family); }
float default_shift(int family) { return family == 2 ? 1.15f : 0.f; }
}  // namespace

REF_API int ref_get_sigmas_sched(int family, int scheduler, int steps, int image_seq_len, float shift, float* out) {
    auto d = make_denoiser_shift(family, shift > 0.f ? shift : default_shift(family));
    const std::vector<float> s = d->get_sigmas((uint32_t)steps, image_seq_len, (scheduler_t)scheduler, version_of4(family));
    std::memcpy(out, s.data(), sizeof(float) * s.size());
    return (int)s.size();
}
// sample_k_diffusion under any method / scheduler on the synthetic model; aux: 5 floats per MODEL CALL in call order (at most aux_calls); returns the number of model calls
REF_API int ref_sample_synthetic2(int family, int steps, int image_seq_len, int64_t n, uint64_t seed, int method, int scheduler, float eta, float* out, float* aux, int aux_calls) {
    auto d = make_denoiser_shift(family, default_shift(family));
    const std::vector<float> sigmas = d->get_sigmas((uint32_t)steps, image_seq_len, (scheduler_t)scheduler, version_of4(family));
    auto rng = std::make_shared<PhiloxRNG>();
    rng->manual_seed(seed);
    sd::Tensor<float> noise  = sd::Tensor<float>::randn({n}, rng);
    sd::Tensor<float> latent = sd::Tensor<float>::zeros({n});
    sd::Tensor<float> x      = d->noise_scaling(sigmas[0], noise, latent);
    int calls = 0;
    denoise_cb_t model = [&](const sd::Tensor<float>& xin, float sigma, int) {
        if (aux && calls < aux_calls) {
            const std::vector<float> sc = d->get_scalings(sigma);
            float* a = aux + 5 * (size_t)calls;
            a[0] = sc[0], a[1] = sc[1], a[2] = sc[2], a[3] = d->sigma_to_t(sigma), a[4] = sigma;
        }
        ++calls;
        sd::guidance::GuiderOutput o;
        o.pred        = sd::Tensor<float>({n});
        const float g = 1.0f / (1.0f + sigma), h = 0.01f * sigma;
        for (int64_t k = 0; k < n; ++k) o.pred.data()[k] = xin.data()[k] * g + h;
        if (method == EULER_CFG_PP_SAMPLE_METHOD || method == EULER_A_CFG_PP_SAMPLE_METHOD) {  // the synthetic model's unconditional prediction
            o.pred_uncond  = sd::Tensor<float>({n});
            const float gu = 0.9f / (1.0f + sigma), hu = -0.02f * sigma;
            for (int64_t k = 0; k < n; ++k) o.pred_uncond.data()[k] = xin.data()[k] * gu + hu;
        }
        return o;
    };
    const bool flow = family == 1 || family == 2;
    sd::Tensor<float> r = sample_k_diffusion((sample_method_t)method, model, std::move(x), sigmas, rng, eta, flow, nullptr);
    if (r.numel() != n) return -1;
    std::memcpy(out, r.data(), sizeof(float) * (size_t)n);
    return calls;
}
