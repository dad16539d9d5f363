// sampler.hpp — host-side fp32 sampler math of the denoise loop.
// Everything here is cheap scalar / elementwise work the reference also keeps on the host.
#pragma once
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
};

// DiscreteFlowDenoiser (src/runtime/denoiser.hpp:1232-1283): SD3 / SD3.5 rectified flow, sigma = shifted t/1000
struct DiscreteFlowDenoiser {
    float shift = 3.0f;
    static float time_snr_shift(float alpha, float t) { return alpha == 1.0f ? t : alpha * t / (1 + (alpha - 1) * t); }
    float sigma_to_t(float sigma) const { return sigma * 1000.f; }
    float t_to_sigma(float t) const { return time_snr_shift(shift, (t + 1) / 1000.f); }
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
