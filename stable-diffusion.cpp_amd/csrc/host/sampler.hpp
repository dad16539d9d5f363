// sampler.hpp — host-side fp32 sampler math of the denoise loop.
// Everything here is cheap scalar / elementwise work the reference also keeps on the host.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace sdmi {

constexpr int TIMESTEPS = 1000;

// Philox4x32-10 + Box-Muller, counter = (offset, 0, i, 0), key = seed — bit-compatible with
// PhiloxRNG::randn (src/core/rng_philox.hpp:101-122), which imitates torch-CUDA randn.
struct PhiloxRNG {
    uint64_t seed   = 0;
    uint32_t offset = 0;
    explicit PhiloxRNG(uint64_t s = 0) : seed(s) {}
    void manual_seed(uint64_t s) {
        seed   = s;
        offset = 0;
    }
    static inline void round(uint32_t c[4], const uint32_t k[2]) {
        const uint64_t p0 = (uint64_t)c[0] * 0xD2511F53u;
        const uint64_t p1 = (uint64_t)c[2] * 0xCD9E8D57u;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0;
        c[1] = n1;
        c[2] = n2;
        c[3] = n3;
    }
    // the integer stage: Philox4x32-10 of counter (offset, 0, i, 0) under key = seed -> 4 words (philox4_32, rng_philox.hpp:63-77)
    static inline void words(uint64_t seed, uint32_t offset, uint32_t i, uint32_t c[4]) {
        c[0] = offset;
        c[1] = 0;
        c[2] = i;
        c[3] = 0;
        uint32_t k[2] = {(uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32)};
        for (int r = 0; r < 9; ++r) {
            round(c, k);
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
        round(c, k);
    }
    std::vector<float> randn(uint32_t n) {
        std::vector<float> out(n);
        const float two_pow32_inv     = 2.3283064e-10f;
        const float two_pow32_inv_2pi = 2.3283064e-10f * 6.2831855f;
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t c[4];
            words(seed, offset, i, c);
            const float u = (float)c[0] * two_pow32_inv + two_pow32_inv / 2;
            const float v = (float)c[1] * two_pow32_inv_2pi + two_pow32_inv_2pi / 2;
            // box_muller (rng_philox.hpp:79-87) calls the UNQUALIFIED log / sqrt / sin with only <cmath> included: under GCC + libstdc++ those
            // are the C double functions (the float overloads live in std::), so the float operands are promoted, the products are
            // formed in double and each statement rounds to float once.  Verified bit-for-bit against the reference's own header
            // compiled from /root/reference (oracle/Makefile -> oracle/_ref/libref_philox.so; tests/golden/philox_ref.npz).
            const float s = (float)std::sqrt((double)-2.0f * std::log((double)u));
            out[i]        = (float)((double)s * std::sin((double)v));
        }
        offset += 1;
        return out;
    }
};

// CompVisDenoiser (src/runtime/denoiser.hpp:1126-1196) with the SD1/SDXL scaled-linear beta table
// (calculate_alphas_cumprod, src/stable-diffusion.cpp:173-186; refresh :666-681)
struct CompVisDenoiser {
    float sigmas[TIMESTEPS];
    float log_sigmas[TIMESTEPS];
    CompVisDenoiser() {
        const float ls_sqrt = sqrtf(0.00085f), le_sqrt = sqrtf(0.0120f);
        const float amount = le_sqrt - ls_sqrt;
        float product      = 1.0f;
        for (int i = 0; i < TIMESTEPS; ++i) {
            const float beta = ls_sqrt + amount * ((float)i / (TIMESTEPS - 1));
            product *= 1.0f - powf(beta, 2.0f);
            sigmas[i]     = std::sqrt((1 - product) / product);
            log_sigmas[i] = std::log(sigmas[i]);
        }
    }
    float sigma_to_t(float sigma) const {
        const float log_sigma = std::log(sigma);
        int low_idx           = 0;
        for (int i = 0; i < TIMESTEPS; ++i)
            if (log_sigma - log_sigmas[i] >= 0) low_idx++;
        low_idx        = std::min(std::max(low_idx - 1, 0), TIMESTEPS - 2);
        const int high = low_idx + 1;
        const float lo = log_sigmas[low_idx], hi = log_sigmas[high];
        float w = (lo - log_sigma) / (lo - hi);
        w       = std::max(0.f, std::min(1.f, w));
        return (1.0f - w) * low_idx + w * high;
    }
    float t_to_sigma(float t) const {
        const int lo = (int)std::floor(t), hi = (int)std::ceil(t);
        const float w = t - (float)lo;
        return std::exp((1.0f - w) * log_sigmas[lo] + w * log_sigmas[hi]);
    }
    // DiscreteScheduler::get_sigmas — denoiser.hpp:32-54
    std::vector<float> get_sigmas(uint32_t n) const {
        std::vector<float> r;
        const int t_max = TIMESTEPS - 1;
        if (n == 0) return r;
        if (n == 1) {
            r.push_back(t_to_sigma((float)t_max));
            r.push_back(0);
            return r;
        }
        const float step = (float)t_max / (float)(n - 1);
        for (uint32_t i = 0; i < n; ++i) r.push_back(t_to_sigma(t_max - step * i));
        r.push_back(0);
        return r;
    }
    // get_scalings — denoiser.hpp:1167-1172: {c_skip, c_out, c_in}
    void scalings(float sigma, float& c_skip, float& c_out, float& c_in) const {
        c_skip = 1.0f;
        c_out  = -sigma;
        c_in   = 1.0f / std::sqrt(sigma * sigma + 1.0f);
    }
    float sigma_min() const { return sigmas[0]; }              // denoiser.hpp:1132-1138
    float sigma_max() const { return sigmas[TIMESTEPS - 1]; }
};

// DiscreteFlowDenoiser (src/runtime/denoiser.hpp:1232-1283): SD3 / SD3.5 rectified flow, sigma = shifted t/1000
struct DiscreteFlowDenoiser {
    float shift = 3.0f;
    static float time_snr_shift(float alpha, float t) { return alpha == 1.0f ? t : alpha * t / (1 + (alpha - 1) * t); }
    float sigma_to_t(float sigma) const { return sigma * 1000.f; }
    float t_to_sigma(float t) const { return time_snr_shift(shift, (t + 1) / 1000.f); }
    float sigma_min() const { return t_to_sigma(0); }  // denoiser.hpp:1250-1256
    float sigma_max() const { return t_to_sigma(TIMESTEPS - 1); }
    std::vector<float> get_sigmas(uint32_t n) const {  // DiscreteScheduler::get_sigmas — denoiser.hpp:32-54
        std::vector<float> r;
        const int t_max = TIMESTEPS - 1;
        if (n == 0) return r;
        if (n == 1) {
            r.push_back(t_to_sigma((float)t_max));
            r.push_back(0);
            return r;
        }
        const float step = (float)t_max / (float)(n - 1);
        for (uint32_t i = 0; i < n; ++i) r.push_back(t_to_sigma(t_max - step * i));
        r.push_back(0);
        return r;
    }
    void scalings(float sigma, float& c_skip, float& c_out, float& c_in) const {
        c_skip = 1.0f;
        c_out  = -sigma;
        c_in   = 1.0f;
    }
};

// FluxFlowDenoiser (denoiser.hpp:1285-1300) with the Flux scheduler (denoiser.hpp:721-782): sigma(t) = e^mu / (e^mu + (1/t - 1)), the
// shift mu interpolated linearly in the image sequence length between (256, 0.5) and (4096, 1.15); the model sees t = sigma
struct FluxFlowDenoiser {
    float base_shift = 0.5f, max_shift = 1.15f;
    static float time_shift(float mu, float sigma, float t) { return ::expf(mu) / (::expf(mu) + ::powf(1.0f / t - 1.0f, sigma)); }
    float sigma_to_t(float sigma) const { return sigma; }
    // schedulers other than the Flux ladder see FluxFlowDenoiser::t_to_sigma (denoiser.hpp:1296-1299): flux_time_shift(shift, 1, (t + 1) / 1000) with the
    // DiscreteFlowDenoiser shift (FLUX.1-dev: 1.15, stable-diffusion.cpp:1822-1827)
    float shift = 1.15f;
    float t_to_sigma(float t) const { return time_shift(shift, 1.0f, (t + 1) / (float)TIMESTEPS); }
    float sigma_min() const { return t_to_sigma(0); }
    float sigma_max() const { return t_to_sigma(TIMESTEPS - 1); }
    std::vector<float> get_sigmas(uint32_t n, int image_seq_len) const {
        const float m  = (max_shift - base_shift) / (4096.0f - 256.0f), b = base_shift - m * 256.0f;
        const float mu = (float)image_seq_len * m + b;
        std::vector<float> s;
        if (n == 0) {
            s.push_back(1.0f);
            return s;
        }
        for (uint32_t i = 0; i <= n; ++i) {
            const float t = 1.0f - (float)i / (float)n;
            s.push_back(t <= 0.0f ? 0.0f : time_shift(mu, 1.0f, t));
        }
        s[n] = 0.0f;
        return s;
    }
    void scalings(float sigma, float& c_skip, float& c_out, float& c_in) const {
        c_skip = 1.0f;
        c_out  = -sigma;
        c_in   = 1.0f;
    }
};

// ---- sigma schedulers other than the family's own ladder (round 6: widening a2 to the reference's scheduler_t) -----------------------------------------
// SigmaScheduler::get_sigmas(n, sigma_min, sigma_max, t_to_sigma) of src/runtime/denoiser.hpp, scheduler by scheduler; numeric values of the enum as in
// include/stable-diffusion.h:65-83.  NOTE on arithmetic: the reference calls log / exp / pow / sqrt UNQUALIFIED on float operands in many places; in its include
// context those are the C double functions (only <cmath>: the float overloads live in std::), so such expressions are formed in double and rounded to float where they
// are stored — restated here operation by operation, and held bit-for-bit against the reference's own header compiled into oracle/_ref (tests/test_host_logic.py).
enum : int {
    SCHED_DISCRETE = 0, SCHED_KARRAS = 1, SCHED_EXPONENTIAL = 2, SCHED_AYS = 3, SCHED_GITS = 4, SCHED_SGM_UNIFORM = 5, SCHED_SIMPLE = 6, SCHED_SMOOTHSTEP = 7,
    SCHED_KL_OPTIMAL = 8, SCHED_LCM = 9, SCHED_BONG_TANGENT = 10, SCHED_LTX2 = 11, SCHED_LOGIT_NORMAL = 12, SCHED_FLUX2 = 13, SCHED_FLUX = 14, SCHED_BETA = 15, SCHED_COUNT = 16
};
inline bool scheduler_supported(int s) {
    return s == SCHED_DISCRETE || s == SCHED_KARRAS || s == SCHED_EXPONENTIAL || s == SCHED_AYS || s == SCHED_SGM_UNIFORM || s == SCHED_SIMPLE || s == SCHED_SMOOTHSTEP ||
           s == SCHED_KL_OPTIMAL || s == SCHED_LCM || s == SCHED_FLUX;
}
// linear_space — denoiser.hpp:122-135 (a running sum, not start + i * inc)
inline std::vector<float> linear_space(float start, float end, size_t num_points) {
    std::vector<float> r(num_points);
    const float inc = (end - start) / (float)(num_points - 1);
    if (num_points > 0) {
        r[0] = start;
        for (size_t i = 1; i < num_points; ++i) r[i] = r[i - 1] + inc;
    }
    return r;
}
// log_linear_interpolation / linear_interp / interp — denoiser.hpp:76-158 (AYS)
inline std::vector<float> log_linear_interpolation(const std::vector<float>& sigma_in, size_t new_len) {
    const size_t s_len = sigma_in.size();
    std::vector<float> x_vals = linear_space(0.f, 1.f, s_len), y_vals(s_len);
    for (size_t i = 0; i < s_len; ++i) y_vals[i] = std::log(sigma_in[s_len - i - 1]);
    std::vector<float> new_x = linear_space(0.f, 1.f, new_len);
    std::vector<double> new_y(new_len);
    if (new_x[0] < x_vals[0]) new_x[0] = x_vals[0];
    if (new_x.back() > x_vals.back()) new_x.back() = x_vals.back();
    size_t i = 0, j = 0;
    while (i < new_len) {
        if (x_vals[j] > new_x[i] || new_x[i] > x_vals[j + 1]) {
            ++j;
            continue;
        }
        const double perc = (double)(new_x[i] - x_vals[j]) / (double)(x_vals[j + 1] - x_vals[j]);
        new_y[i]          = ((double)y_vals[j] * (1. - perc)) + ((double)y_vals[j + 1] * perc);
        ++i;
    }
    std::vector<float> out(new_len);
    for (size_t k = 0; k < new_len; ++k) out[k] = (float)std::exp(new_y[new_len - k - 1]);
    return out;
}
// ays_version: 0 = SD1.x table, 1 = SDXL table (AYSScheduler, denoiser.hpp:163-215); anything else is "not compatible" there (n + 1 zeros)
template <class TToSigma>
inline std::vector<float> scheduler_sigmas(int sched, uint32_t n, float sigma_min, float sigma_max, TToSigma&& t_to_sigma, int ays_version) {
    std::vector<float> r;
    const int t_max = TIMESTEPS - 1;
    switch (sched) {
        case SCHED_KARRAS: {  // denoiser.hpp:285-306
            const float rho = 7.f;
            if (sigma_min <= 1e-6f) sigma_min = 1e-6f;
            r.assign(n + 1, 0.f);
            const float min_inv_rho = (float)pow((double)sigma_min, (double)(1.f / rho));
            const float max_inv_rho = (float)pow((double)sigma_max, (double)(1.f / rho));
            for (uint32_t i = 0; i < n; ++i) r[i] = (float)pow((double)(max_inv_rho + (float)i / ((float)n - 1.f) * (min_inv_rho - max_inv_rho)), (double)rho);
            r[n] = 0.f;
            return r;
        }
        case SCHED_EXPONENTIAL: {  // denoiser.hpp:56-76
            const float lmin = std::log(sigma_min), lmax = std::log(sigma_max);
            const float step = (lmax - lmin) / (float)(n - 1);
            for (uint32_t i = 0; i < n; ++i) r.push_back(std::exp(lmax - step * (float)i));
            r.push_back(0.f);
            return r;
        }
        case SCHED_AYS: {
            static const float sd15[11] = {14.6146412293f, 6.4745760956f, 3.8636745985f, 2.6946151520f, 1.8841921177f, 1.3943805092f, 0.9642583904f, 0.6523686016f, 0.3977456272f, 0.1515232662f, 0.0291671582f};
            static const float sdxl[11] = {14.6146412293f, 6.3184485287f, 3.7681790315f, 2.1811480769f, 1.3405244945f, 0.8620721141f, 0.5550693289f, 0.3798540708f, 0.2332364134f, 0.1114188177f, 0.0291671582f};
            r.assign(n + 1, 0.f);
            if (ays_version != 0 && ays_version != 1) return r;
            const std::vector<float> inputs(ays_version == 0 ? sd15 : sdxl, (ays_version == 0 ? sd15 : sdxl) + 11);
            r = (n + 1 != inputs.size()) ? log_linear_interpolation(inputs, n + 1) : inputs;
            r[n] = 0.f;
            return r;
        }
        case SCHED_SGM_UNIFORM: {  // denoiser.hpp:249-266
            if (n == 0) return {0.f};
            const std::vector<float> ts = linear_space((float)t_max, 0.f, n + 1);
            for (uint32_t i = 0; i < n; ++i) r.push_back(t_to_sigma(ts[i]));
            r.push_back(0.f);
            return r;
        }
        case SCHED_SIMPLE: {  // denoiser.hpp:464-491
            if (n == 0) return r;
            const float step_factor = (float)TIMESTEPS / (float)n;
            for (uint32_t i = 0; i < n; ++i) {
                int idx = TIMESTEPS - 1 - (int)((float)i * step_factor);
                if (idx < 0) idx = 0;
                r.push_back(t_to_sigma((float)idx));
            }
            r.push_back(0.f);
            return r;
        }
        case SCHED_SMOOTHSTEP: {  // denoiser.hpp:494-520
            if (n == 0) return r;
            if (n == 1) return {t_to_sigma((float)t_max), 0.f};
            for (uint32_t i = 0; i < n; ++i) {
                const float u = 1.f - (float)i / (float)n;
                r.push_back(t_to_sigma(std::round(u * u * (3.0f - 2.0f * u) * t_max)));
            }
            r.push_back(0.f);
            return r;
        }
        case SCHED_KL_OPTIMAL: {  // denoiser.hpp:611-644
            if (n == 0) return r;
            if (n == 1) return {sigma_max, 0.f};
            if (sigma_min <= 1e-6f) sigma_min = 1e-6f;
            const float amin = std::atan(sigma_min), amax = std::atan(sigma_max);
            for (uint32_t i = 0; i < n; ++i) {
                const float t = (float)i / (float)(n - 1);
                r.push_back(std::tan(t * amin + (1.0f - t) * amax));
            }
            r.push_back(0.f);
            return r;
        }
        case SCHED_LCM: {  // denoiser.hpp:268-283
            const int original_steps = 50, k = TIMESTEPS / original_steps;
            for (uint32_t i = 0; i < n; ++i) {
                const int index = (int)((i * (uint32_t)original_steps) / n);
                r.push_back(t_to_sigma((float)((original_steps - index) * k - 1)));
            }
            r.push_back(0.f);
            return r;
        }
        default: {  // DiscreteScheduler — denoiser.hpp:32-54
            if (n == 0) return r;
            if (n == 1) return {t_to_sigma((float)t_max), 0.f};
            const float step = (float)t_max / (float)(n - 1);
            for (uint32_t i = 0; i < n; ++i) r.push_back(t_to_sigma(t_max - step * i));
            r.push_back(0.f);
            return r;
        }
    }
}

// get_ancestral_step — denoiser.hpp:1447-1467
inline void ancestral_step(float sigma_from, float sigma_to, float eta, float& sigma_down, float& sigma_up) {
    sigma_up   = 0.0f;
    sigma_down = sigma_to;
    if (eta <= 0.0f) return;
    const float from_sq = sigma_from * sigma_from, to_sq = sigma_to * sigma_to;
    if (from_sq > 0.0f) {
        const float term = to_sq * (from_sq - to_sq) / from_sq;
        sigma_up         = std::min(sigma_to, eta * std::sqrt(std::max(term, 0.0f)));
    }
    const float down_sq = to_sq - sigma_up * sigma_up;
    sigma_down          = down_sq > 0.0f ? std::sqrt(down_sq) : 0.0f;
}

// get_ancestral_step_flow — denoiser.hpp:1468-1499 (rectified-flow denoisers: SD3 / SD3.5 / FLUX).  eta is clamped to 1; the caller scales x by
// alpha_scale before adding sigma_up * noise (sample_euler_ancestral, denoiser.hpp:1536-1541)
inline void ancestral_step_flow(float sigma_from, float sigma_to, float eta, float& sigma_down, float& sigma_up, float& alpha_scale) {
    sigma_down  = sigma_to;
    sigma_up    = 0.0f;
    alpha_scale = 1.0f;
    if (eta <= 0.0f || sigma_from <= 0.0f || sigma_to <= 0.0f) return;
    eta                     = std::min(eta, 1.0f);
    const float sigma_ratio = sigma_to / sigma_from;
    sigma_down              = sigma_to * (1.0f + (sigma_ratio - 1.0f) * eta);
    sigma_down              = std::max(0.0f, std::min(sigma_to, sigma_down));
    const float denom       = 1.0f - sigma_down;
    if (denom <= 0.0f) {
        sigma_down = sigma_to;
        return;
    }
    alpha_scale = (1.0f - sigma_to) / denom;
    float term  = (sigma_down / sigma_to) * alpha_scale;
    term        = std::max(-1.0f, std::min(1.0f, term));
    sigma_up    = sigma_to * std::sqrt(std::max(1.0f - term * term, 0.0f));
}


// ---- the reference's other k-diffusion samplers (round 6: widening a2 to sample_method_t) -------------------------------------------------------------------
// sample_heun / sample_dpm2 / sample_dpmpp_2s_ancestral[_flow] / sample_dpmpp_2m / sample_dpmpp_2m_v2 / sample_ipndm / sample_ipndm_v / sample_lcm
// (src/runtime/denoiser.hpp:1599-1860, 2059-2228) on flat float arrays: every sd::Tensor<float> operator of the reference is one loop here, with the scalar cast to
// float before it meets the data exactly as tensor.hpp:610-650, 745-800 does; scalar expressions follow the reference's own types (unqualified log / exp = the C double
// functions, see the scheduler note above).  Numeric values of the methods as in include/stable-diffusion.h:38-61.  Held bit-for-bit against the reference's own header
// compiled into oracle/_ref on whole trajectories (tests/test_host_logic.py::test_more_samplers_bit_exact_vs_reference).
enum : int {
    SM_EULER = 0, SM_EULER_A = 1, SM_HEUN = 2, SM_DPM2 = 3, SM_DPMPP2S_A = 4, SM_DPMPP2M = 5, SM_DPMPP2Mv2 = 6, SM_IPNDM = 7, SM_IPNDM_V = 8, SM_LCM = 9, SM_DDIM_TRAILING = 10,
    SM_COUNT = 21
};
inline bool sample_method_supported(int m) { return m >= SM_EULER && m <= SM_DDIM_TRAILING; }
// resolve_eta — src/stable-diffusion.cpp:4024-4049
inline float default_eta(int method) { return (method == SM_EULER_A || method == SM_DPMPP2S_A) ? 1.0f : 0.0f; }

// model(x, sigma, denoised) -> false on failure; randn(out): one N(0,1) draw per element of x from the sampler's RNG stream(s) (sd::Tensor<float>::randn_like(x, rng))
template <class ModelFn, class RandnFn>
inline bool run_sampler_generic(int method, ModelFn&& model, std::vector<float>& x, const std::vector<float>& sigmas, RandnFn&& randn, float eta, bool flow) {
    const size_t n  = x.size();
    const int steps = (int)sigmas.size() - 1;
    std::vector<float> den(n), x2(n), den2(n), d(n), nz;
    auto t_fn     = [](float sigma) -> float { return (float)(-log((double)sigma)); };
    auto sigma_fn = [](float t) -> float { return (float)exp((double)-t); };
    switch (method) {
        case SM_HEUN:
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data())) return false;
                const float dt = sigmas[i + 1] - sigmas[i];
                for (size_t k = 0; k < n; ++k) d[k] = (x[k] - den[k]) / sigmas[i];
                if (sigmas[i + 1] == 0) {
                    for (size_t k = 0; k < n; ++k) x[k] += d[k] * dt;
                } else {
                    for (size_t k = 0; k < n; ++k) x2[k] = x[k] + d[k] * dt;
                    if (!model(x2.data(), sigmas[i + 1], den2.data())) return false;
                    for (size_t k = 0; k < n; ++k) {
                        const float d2 = (x2[k] - den2[k]) / sigmas[i + 1];
                        const float dm = (d[k] + d2) / 2.0f;
                        x[k] += dm * dt;
                    }
                }
            }
            return true;
        case SM_DPM2:
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data())) return false;
                for (size_t k = 0; k < n; ++k) d[k] = (x[k] - den[k]) / sigmas[i];
                if (sigmas[i + 1] == 0) {
                    const float dt = sigmas[i + 1] - sigmas[i];
                    for (size_t k = 0; k < n; ++k) x[k] += d[k] * dt;
                } else {
                    const float sigma_mid = (float)exp(0.5f * (log((double)sigmas[i]) + log((double)sigmas[i + 1])));
                    const float dt_1 = sigma_mid - sigmas[i], dt_2 = sigmas[i + 1] - sigmas[i];
                    for (size_t k = 0; k < n; ++k) x2[k] = x[k] + d[k] * dt_1;
                    if (!model(x2.data(), sigma_mid, den2.data())) return false;
                    for (size_t k = 0; k < n; ++k) {
                        const float d2 = (x2[k] - den2[k]) / sigma_mid;
                        x[k] += d2 * dt_2;
                    }
                }
            }
            return true;
        case SM_DPMPP2S_A:
            if (!flow) {
                for (int i = 0; i < steps; ++i) {
                    if (!model(x.data(), sigmas[i], den.data())) return false;
                    float sigma_down, sigma_up;
                    ancestral_step(sigmas[i], sigmas[i + 1], eta, sigma_down, sigma_up);
                    if (sigma_down == 0) {
                        x = den;
                    } else {
                        const float t = t_fn(sigmas[i]), t_next = t_fn(sigma_down), h = t_next - t, s = t + 0.5f * h, sigma_s = sigma_fn(s);
                        const float a1 = sigma_s / sigma_fn(t), b1 = (float)(exp((double)(-h * 0.5f)) - 1);
                        for (size_t k = 0; k < n; ++k) x2[k] = a1 * x[k] - b1 * den[k];
                        if (!model(x2.data(), sigma_s, den2.data())) return false;
                        const float a2 = sigma_fn(t_next) / sigma_fn(t), b2 = (float)(exp((double)-h) - 1);
                        for (size_t k = 0; k < n; ++k) x[k] = a2 * x[k] - b2 * den2[k];
                    }
                    if (sigmas[i + 1] > 0) {
                        nz.resize(n);
                        randn(nz.data());
                        for (size_t k = 0; k < n; ++k) x[k] += nz[k] * sigma_up;
                    }
                }
                return true;
            }
            for (int i = 0; i < steps; ++i) {  // sample_dpmpp_2s_ancestral_flow, denoiser.hpp:1697-1789
                const float sigma = sigmas[i], sigma_to = sigmas[i + 1];
                const bool opt_first_step = (1.0 - (double)sigma < 1e-6);
                if (!model(x.data(), sigma, den.data())) return false;
                if (sigma_to == 0.0f) {
                    x = den;
                    continue;
                }
                float sigma_down, sigma_up, alpha_scale;
                ancestral_step_flow(sigma, sigma_to, eta, sigma_down, sigma_up, alpha_scale);
                const float* D_i = den.data();
                if (!opt_first_step) {
                    const float exp_s = std::sqrt(((1 - sigma) / sigma) * ((1 - sigma_down) / sigma_down));
                    const float sigma_s = 1.0f / (exp_s + 1.0f), ratio = sigma_s / sigma, omr = 1.0f - ratio;
                    for (size_t k = 0; k < n; ++k) x2[k] = (x[k] * ratio) + (den[k] * omr);
                    if (!model(x2.data(), sigma_s, den2.data())) return false;
                    D_i = den2.data();
                }
                const float rd = sigma_down / sigma, omrd = 1.0f - rd;
                for (size_t k = 0; k < n; ++k) x[k] = (x[k] * rd) + (D_i[k] * omrd);
                if (sigma_to > 0.0f && eta > 0.0f) {
                    nz.resize(n);
                    randn(nz.data());
                    for (size_t k = 0; k < n; ++k) x[k] = alpha_scale * x[k] + nz[k] * sigma_up;
                }
            }
            return true;
        case SM_DPMPP2M:
        case SM_DPMPP2Mv2: {
            std::vector<float> old = x;
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data())) return false;
                const float t = t_fn(sigmas[i]), t_next = t_fn(sigmas[i + 1]), h = t_next - t, a = sigmas[i + 1] / sigmas[i];
                if (i == 0 || sigmas[i + 1] == 0) {
                    const float b = (float)(exp((double)-h) - 1.f);
                    for (size_t k = 0; k < n; ++k) x[k] = a * x[k] - b * den[k];
                } else {
                    const float h_last = t - t_fn(sigmas[i - 1]);
                    float r, b;
                    if (method == SM_DPMPP2M) {
                        r = h_last / h;
                        b = (float)(exp((double)-h) - 1.f);
                    } else {
                        const float h_min = std::min(h_last, h), h_max = std::max(h_last, h);
                        r                 = h_max / h_min;
                        const float h_d   = (h_max + h_min) / 2.f;
                        b                 = (float)(exp((double)-h_d) - 1.f);
                    }
                    const float c1 = 1.f + 1.f / (2.f * r), c2 = 1.f / (2.f * r);
                    for (size_t k = 0; k < n; ++k) {
                        const float dd = c1 * den[k] - c2 * old[k];
                        x[k]           = a * x[k] - b * dd;
                    }
                }
                old = den;
            }
            return true;
        }
        case SM_IPNDM:
        case SM_IPNDM_V: {
            const int max_order = 4;
            std::vector<std::vector<float>> hist;
            for (int i = 0; i < steps; ++i) {
                const float sigma = sigmas[i], sigma_next = sigmas[i + 1];
                if (!model(x.data(), sigma, den.data())) return false;
                std::vector<float> dc(n);
                for (size_t k = 0; k < n; ++k) dc[k] = (x[k] - den[k]) / sigma;
                const int order = std::min(max_order, i + 1);
                const float dt  = sigma_next - sigma;
                const float hn1 = (i > 0) ? (sigma - sigmas[i - 1]) : dt;
                const size_t hs = hist.size();
                switch (order) {
                    case 1:
                        for (size_t k = 0; k < n; ++k) x[k] += dc[k] * dt;
                        break;
                    case 2:
                        if (method == SM_IPNDM) {
                            for (size_t k = 0; k < n; ++k) x[k] += ((3.f * dc[k] - hist[hs - 1][k]) / 2.f) * dt;
                        } else {
                            const float q = dt / hn1, c = 2.f + q;
                            for (size_t k = 0; k < n; ++k) x[k] += ((c * dc[k] - q * hist[hs - 1][k]) / 2.f) * dt;
                        }
                        break;
                    case 3:
                        for (size_t k = 0; k < n; ++k) x[k] += ((23.f * dc[k] - 16.f * hist[hs - 1][k] + 5.f * hist[hs - 2][k]) / 12.f) * dt;
                        break;
                    default:
                        for (size_t k = 0; k < n; ++k) x[k] += ((55.f * dc[k] - 59.f * hist[hs - 1][k] + 37.f * hist[hs - 2][k] - 9.f * hist[hs - 3][k]) / 24.f) * dt;
                        break;
                }
                if (hist.size() == (size_t)(max_order - 1)) hist.erase(hist.begin());
                hist.push_back(std::move(dc));
            }
            return true;
        }
        case SM_LCM:
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data())) return false;
                x = den;
                if (sigmas[i + 1] > 0) {
                    if (flow) {
                        const float f = 1 - sigmas[i + 1];
                        for (size_t k = 0; k < n; ++k) x[k] *= f;
                    }
                    nz.resize(n);
                    randn(nz.data());
                    const float t = steps > 1 ? (float)i / (float)(steps - 1) : 0.0f;
                    const float noise_scale = 1.0f + (1.0f - 1.0f) * t;  // noise_scale_start / _end defaults (no extra sample args)
                    const float f2 = sigmas[i + 1] * noise_scale;
                    for (size_t k = 0; k < n; ++k) x[k] += nz[k] * f2;
                }
            }
            return true;
        default:
            return false;
    }
}

}  // namespace sdmi

// classifier-free guidance on one element — sd::guidance::ClassifierFreeGuidance::forward, src/runtime/guidance.cpp:171: pred_uncond + guidance_scale *
// (pred_cond - pred_uncond) on sd::Tensor<float>, i.e. THREE separately rounded f32 operations (difference, scaled difference, sum).  The host library is
// built without FMA contraction (build.py HOST_FLAGS: baseline x86-64), and the volatile steps keep it that way under any flags: bit-exact against the
// reference's own code (tests/test_host_logic.py::test_cfg_combine_bit_exact_vs_reference).
inline float cfg_guided(float cond, float uncond, float scale) {
    volatile float d = cond - uncond;
    volatile float s = scale * d;
    return uncond + s;
}

// One sampler update on `nb` images of `per` floats each — the arithmetic of the reference's sd::Tensor<float> expressions, operation by operation (every
// tensor operator rounds to f32; scalars are cast to float before they meet the tensor, src/core/tensor.hpp:612-618, 750-760):
//   Euler-A  (sample_euler_ancestral, src/runtime/denoiser.hpp:1513-1546):  sigma_to == 0: x = denoised;
//            eta == 0: x = r*x + (1 - r)*denoised with r = sigma_to / sigma;  else r = sigma_down / sigma, the same blend, then (sigma_up > 0)
//            [flow denoisers: x *= alpha_scale], x += noise * sigma_up
//   Euler    (sample_euler, :1582-1597):  d = (x - denoised) / sigma;  x += d * (sigma_to - sigma)
// noise(b) returns image b's `per` ancestral-noise floats (asked for only when sigma_up > 0).  The host library is built for baseline x86-64 (no FMA
// contraction); tests/test_host_logic.py holds whole trajectories of this function bit-for-bit against the reference's own code (oracle/_ref).
template <class NoiseFn>
inline void sampler_update(float* x, const float* denoised, size_t per, int nb, bool euler_a, bool flow, float sigma, float sigma_to, float eta, float sigma_down,
                           float sigma_up, float alpha_scale, NoiseFn&& noise) {
    const size_t n = per * (size_t)nb;
    if (!euler_a) {
        const float ds = sigma_to - sigma;
        for (size_t k = 0; k < n; ++k) {
            const float d = (x[k] - denoised[k]) / sigma;
            x[k] += d * ds;
        }
        return;
    }
    if (sigma_to == 0.f) {
        for (size_t k = 0; k < n; ++k) x[k] = denoised[k];
        return;
    }
    const float ratio = (eta == 0.f ? sigma_to : sigma_down) / sigma;
    const float one_m = 1.0f - ratio;  // eta == 0: the reference writes (1.0 - ratio) in double and casts the scalar to float: the same value
    for (size_t k = 0; k < n; ++k) x[k] = ratio * x[k] + one_m * denoised[k];
    if (eta != 0.f && sigma_up > 0.f) {
        if (flow)
            for (size_t k = 0; k < n; ++k) x[k] *= alpha_scale;
        for (int b = 0; b < nb; ++b) {
            const float* nz = noise(b);
            for (size_t k = 0; k < per; ++k) x[(size_t)b * per + k] += nz[k] * sigma_up;
        }
    }
}
