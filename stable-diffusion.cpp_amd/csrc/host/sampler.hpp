// sampler.hpp — host-side fp32 sampler math of the denoise loop.
// Everything here is cheap scalar / elementwise work the reference also keeps on the host.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <random>
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
           s == SCHED_KL_OPTIMAL || s == SCHED_LCM || s == SCHED_FLUX || s == SCHED_GITS || s == SCHED_BONG_TANGENT || s == SCHED_BETA;
}
// (not implemented: LTX2, logit-normal and FLUX.2 — the default ladders of model families outside this engine: LTX video, Ideogram, FLUX.2; include/sd-mi355x.h)

// BetaScheduler's quantile function (denoiser.hpp:308-463) with its default alpha = beta = 0.6: regularised incomplete beta by Lentz's continued fraction, Newton steps
struct BetaQuantile {
    static double log_beta(double a, double b) { return std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b); }
    // I_x(a, b) by the modified Lentz evaluation of its continued fraction: every partial numerator `term` updates the pair (D, C) and contributes the factor D * C;
    // numerators alternate between the even form m (b - m) x / ((a + 2m - 1)(a + 2m)) and the odd form -(a + m)(a + b + m) x / ((a + 2m)(a + 2m + 1)).  Operation order as
    // the reference evaluates it (the ladders are compared bit for bit).
    static double incbeta(double x, double a, double b) {
        if (x <= 0.0) return 0.0;
        if (x >= 1.0) return 1.0;
        const double tiny = 1e-30, tol = 3.0e-7;
        const double apb = a + b, ap1 = a + 1.0, am1 = a - 1.0;
        double C = 1.0, D = 1.0 - apb * x / ap1;
        if (std::abs(D) < tiny) D = tiny;
        D          = 1.0 / D;
        double frac = D;
        auto factor = [&](double term) {
            D = 1.0 + term * D;
            if (std::abs(D) < tiny) D = tiny;
            C = 1.0 + term / C;
            if (std::abs(C) < tiny) C = tiny;
            D = 1.0 / D;
            return D * C;
        };
        for (int m = 1; m <= 200; m++) {
            const int m2 = 2 * m;
            frac *= factor(m * (b - m) * x / ((am1 + m2) * (a + m2)));
            const double last = factor(-(a + m) * (apb + m) * x / ((a + m2) * (ap1 + m2)));
            frac *= last;
            if (std::abs(last - 1.0) < tol) break;
        }
        return std::exp(a * std::log(x) + b * std::log(1.0 - x) - log_beta(a, b)) / a * frac;
    }
    static double cdf(double x, double a, double b) {
        if (x == 0.0) return 0.0;
        if (x == 1.0) return 1.0;
        if (x < (a + 1.0) / (a + b + 2.0)) return incbeta(x, a, b);
        return 1.0 - incbeta(1.0 - x, b, a);
    }
    static double ppf(double u, double a, double b, int max_iter = 30) {
        double x = 0.5;
        for (int i = 0; i < max_iter; i++) {
            const double f = cdf(x, a, b) - u;
            if (std::abs(f) < 1e-10) break;
            const double df = std::exp((a - 1.0) * std::log(x) + (b - 1.0) * std::log(1.0 - x) - log_beta(a, b));
            x -= f / df;
            if (x <= 0.0) x = 1e-10;
            if (x >= 1.0) x = 1.0 - 1e-10;
        }
        return x;
    }
};
// BongTangentScheduler::get_bong_tangent_sigmas — denoiser.hpp:525-558
inline std::vector<float> bong_tangent_sigmas(int steps, float slope, float pivot, float start, float end) {
    const float kPi = 3.14159265358979323846f;
    std::vector<float> sigmas;
    if (steps <= 0) return sigmas;
    const float smax   = ((2.0f / kPi) * atanf(-slope * (0.0f - pivot)) + 1.0f) * 0.5f;
    const float smin   = ((2.0f / kPi) * atanf(-slope * ((float)(steps - 1) - pivot)) + 1.0f) * 0.5f;
    const float srange = smax - smin, sscale = start - end;
    if (fabsf(srange) < 1e-8f) {
        if (steps == 1) return {start};
        for (int i = 0; i < steps; ++i) {
            const float t = (float)i / (float)(steps - 1);
            sigmas.push_back(start + (end - start) * t);
        }
        return sigmas;
    }
    const float inv_srange = 1.0f / srange;
    for (int x = 0; x < steps; ++x) {
        const float v = ((2.0f / kPi) * atanf(-slope * ((float)x - pivot)) + 1.0f) * 0.5f;
        sigmas.push_back(((v - smin) * inv_srange) * sscale + end);
    }
    return sigmas;
}
// GITS (https://github.com/zju-pi/diff-sampler, "gits-main"): the published noise ladders for coefficient 1.20 — the one GITSScheduler always selects (denoiser.hpp:220-247:
// coeff is fixed at 1.20f -> index 8 of GITS_NOISE) — for 2 ... 20 steps, row n - 2 holding n + 1 sigmas
inline const float* gits_noise_1_20(uint32_t n) {
    static const float T[] = {
        14.61464119f, 0.803307f, 0.02916753f,
        14.61464119f, 1.56271636f, 0.52423614f, 0.02916753f,
        14.61464119f, 2.36326075f, 0.92192322f, 0.36617002f, 0.02916753f,
        14.61464119f, 2.84484982f, 1.24153244f, 0.59516323f, 0.25053367f, 0.02916753f,
        14.61464119f, 5.85520077f, 2.05039096f, 0.95350921f, 0.45573691f, 0.17026083f, 0.02916753f,
        14.61464119f, 5.85520077f, 2.45070267f, 1.24153244f, 0.64427125f, 0.29807833f, 0.09824532f, 0.02916753f,
        14.61464119f, 5.85520077f, 2.45070267f, 1.36964464f, 0.803307f, 0.45573691f, 0.25053367f, 0.09824532f, 0.02916753f,
        14.61464119f, 5.85520077f, 2.84484982f, 1.61558151f, 0.95350921f, 0.59516323f, 0.36617002f, 0.19894916f, 0.09824532f, 0.02916753f,
        14.61464119f, 5.85520077f, 2.84484982f, 1.67050016f, 1.08895338f, 0.74807048f, 0.50118381f, 0.32104823f, 0.19894916f, 0.09824532f, 0.02916753f,
        14.61464119f, 5.85520077f, 2.95596409f, 1.84880662f, 1.24153244f, 0.83188516f, 0.59516323f, 0.41087446f, 0.27464288f, 0.17026083f, 0.09824532f, 0.02916753f,
        14.61464119f, 5.85520077f, 3.07277966f, 1.98035145f, 1.36964464f, 0.95350921f, 0.69515091f, 0.50118381f, 0.36617002f, 0.25053367f, 0.17026083f, 0.09824532f, 0.02916753f,
        14.61464119f, 6.77309084f, 3.46139455f, 2.36326075f, 1.56271636f, 1.08895338f, 0.803307f, 0.59516323f, 0.45573691f, 0.34370604f, 0.25053367f, 0.17026083f, 0.09824532f,
        0.02916753f,
        14.61464119f, 6.77309084f, 3.46139455f, 2.45070267f, 1.61558151f, 1.162866f, 0.86115354f, 0.64427125f, 0.50118381f, 0.38853383f, 0.29807833f, 0.22545385f, 0.17026083f,
        0.09824532f, 0.02916753f,
        14.61464119f, 7.49001646f, 4.65472794f, 3.07277966f, 2.12350607f, 1.51179266f, 1.08895338f, 0.83188516f, 0.64427125f, 0.50118381f, 0.38853383f, 0.29807833f, 0.22545385f,
        0.17026083f, 0.09824532f, 0.02916753f,
        14.61464119f, 7.49001646f, 4.65472794f, 3.07277966f, 2.12350607f, 1.51179266f, 1.08895338f, 0.83188516f, 0.64427125f, 0.50118381f, 0.41087446f, 0.32104823f, 0.25053367f,
        0.19894916f, 0.13792117f, 0.09824532f, 0.02916753f,
        14.61464119f, 7.49001646f, 4.65472794f, 3.07277966f, 2.12350607f, 1.51179266f, 1.08895338f, 0.83188516f, 0.64427125f, 0.50118381f, 0.41087446f, 0.34370604f, 0.27464288f,
        0.22545385f, 0.17026083f, 0.13792117f, 0.09824532f, 0.02916753f,
        14.61464119f, 7.49001646f, 4.65472794f, 3.07277966f, 2.19988537f, 1.61558151f, 1.20157266f, 0.92192322f, 0.72133851f, 0.57119018f, 0.45573691f, 0.36617002f, 0.29807833f,
        0.25053367f, 0.19894916f, 0.17026083f, 0.13792117f, 0.09824532f, 0.02916753f,
        14.61464119f, 7.49001646f, 4.65472794f, 3.07277966f, 2.19988537f, 1.61558151f, 1.24153244f, 0.95350921f, 0.74807048f, 0.59516323f, 0.4783645f, 0.38853383f, 0.32104823f,
        0.27464288f, 0.22545385f, 0.19894916f, 0.17026083f, 0.13792117f, 0.09824532f, 0.02916753f,
        14.61464119f, 7.49001646f, 4.65472794f, 3.07277966f, 2.19988537f, 1.61558151f, 1.24153244f, 0.95350921f, 0.74807048f, 0.59516323f, 0.50118381f, 0.41087446f, 0.34370604f,
        0.29807833f, 0.25053367f, 0.22545385f, 0.19894916f, 0.17026083f, 0.13792117f, 0.09824532f, 0.02916753f};
    static_assert(sizeof(T) / sizeof(T[0]) == 228, "19 ladders of 3 ... 21 sigmas");
    size_t off = 0;
    for (uint32_t k = 2; k < n; ++k) off += k + 1;
    return T + off;
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
        case SCHED_GITS: {  // denoiser.hpp:220-247 (n < 2 indexes in front of the table there: no ladder here)
            if (sigma_max <= 0.0f || n < 2) return r;
            if (n <= 20) {
                const float* row = gits_noise_1_20(n);
                r.assign(row, row + n + 1);
            } else {
                const float* last = gits_noise_1_20(20);
                r                 = log_linear_interpolation(std::vector<float>(last, last + 21), n + 1);
            }
            r[n] = 0.0f;
            return r;
        }
        case SCHED_BETA: {  // denoiser.hpp:434-462, alpha = beta = 0.6 (no extra sample args); repeated timesteps are dropped: the ladder may be SHORTER than n + 1
            if (n == 0) return r;
            if (n == 1) return {t_to_sigma((float)t_max), 0.f};
            int last_t = -1;
            for (uint32_t i = 0; i < n; i++) {
                const double u      = 1.0 - static_cast<double>(i) / static_cast<double>(n);
                const double t_cont = BetaQuantile::ppf(u, 0.6, 0.6) * t_max;
                const int t         = static_cast<int>(std::lround(t_cont));
                if (t != last_t) {
                    r.push_back(t_to_sigma(static_cast<float>(t)));
                    last_t = t;
                }
            }
            r.push_back(0.f);
            return r;
        }
        case SCHED_BONG_TANGENT: {  // denoiser.hpp:560-608
            if (n == 0) return r;
            const float start = sigma_max, end = sigma_min, middle = sigma_min + (sigma_max - sigma_min) * 0.5f;
            const float pivot_1 = 0.6f, pivot_2 = 0.6f;
            float slope_1 = 0.2f, slope_2 = 0.2f;
            const int steps = static_cast<int>(n) + 2;
            const int midpoint  = static_cast<int>(((float)steps * pivot_1 + (float)steps * pivot_2) * 0.5f);
            const int pivot_1_i = static_cast<int>((float)steps * pivot_1), pivot_2_i = static_cast<int>((float)steps * pivot_2);
            const float slope_scale = (float)steps / 40.0f;
            slope_1 = slope_1 / slope_scale;
            slope_2 = slope_2 / slope_scale;
            const int stage_2_len = steps - midpoint, stage_1_len = steps - stage_2_len;
            std::vector<float> s1       = bong_tangent_sigmas(stage_1_len, slope_1, (float)pivot_1_i, start, middle);
            const std::vector<float> s2 = bong_tangent_sigmas(stage_2_len, slope_2, (float)(pivot_2_i - stage_1_len), middle, end);
            if (!s1.empty()) s1.pop_back();
            r.insert(r.end(), s1.begin(), s1.end());
            r.insert(r.end(), s2.begin(), s2.end());
            if (r.size() < n + 1) {
                while (r.size() < n + 1) r.push_back(end);
            } else if (r.size() > n + 1) {
                r.resize(n + 1);
            }
            r[n] = 0.0f;
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
    SM_TCD = 11, SM_RES_MULTISTEP = 12, SM_RES_2S = 13, SM_ER_SDE = 14, SM_EULER_CFG_PP = 15, SM_EULER_A_CFG_PP = 16, SM_EULER_GE = 17, SM_DPMPP2M_SDE = 18,
    SM_DPMPP2M_SDE_BT = 19, SM_LMS = 20, SM_COUNT = 21
};
inline bool sample_method_supported(int m) { return m >= SM_EULER && m < SM_COUNT; }
// the CFG++ methods step along the UNCONDITIONAL prediction: the model callback is asked for it (GuiderOutput::pred_uncond, stable-diffusion.cpp:2877-2884)
inline bool sample_method_needs_uncond(int m) { return m == SM_EULER_CFG_PP || m == SM_EULER_A_CFG_PP; }
// resolve_eta — src/stable-diffusion.cpp:4024-4049
inline float default_eta(int method) {
    switch (method) {
        case SM_EULER_A: case SM_DPMPP2S_A: case SM_ER_SDE: case SM_EULER_A_CFG_PP: case SM_DPMPP2M_SDE: case SM_DPMPP2M_SDE_BT: return 1.0f;
        default: return 0.0f;
    }
}

// get_ancestral_step(sigma_from, sigma_to, eta, is_flow_denoiser) — denoiser.hpp:1501-1511
inline void ancestral_step3(float sigma_from, float sigma_to, float eta, bool flow, float& sigma_down, float& sigma_up, float& alpha_scale) {
    if (flow) {
        ancestral_step_flow(sigma_from, sigma_to, eta, sigma_down, sigma_up, alpha_scale);
    } else {
        ancestral_step(sigma_from, sigma_to, eta, sigma_down, sigma_up);
        alpha_scale = 1.0f;
    }
}

// STDDefaultRNG (src/core/rng.hpp:14-33): std::default_random_engine seeded with the low 32 bits, a FRESH std::normal_distribution<float> per call — the generator the
// reference's Brownian tree draws from.  (Both sides are libstdc++ here; the stream is that library's minstd_rand0 + Marsaglia polar method.)
inline std::vector<float> std_default_randn(uint64_t seed, size_t n) {
    std::default_random_engine generator;
    generator.seed((unsigned int)seed);
    std::normal_distribution<float> distribution(0.0f, 1.0f);
    std::vector<float> r;
    r.reserve(n);
    for (size_t i = 0; i < n; ++i) r.push_back(distribution(generator));
    return r;
}

// BrownianTreeNoiseSampler (src/runtime/denoiser.hpp:1909-1993): deterministic Brownian increments over [sigma_min, sigma_max] for ONE image of n floats
struct BrownianTree {
    static constexpr int kMaxDepth = 24;
    double t_min, t_max;
    size_t n;
    uint64_t root_seed;
    std::vector<float> w_at_tmax;
    std::map<double, std::vector<float>> cache;
    static uint64_t mix64(uint64_t v, uint64_t salt) {
        uint64_t z = v + salt;
        z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z          = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    BrownianTree(size_t n_, double sigma_min, double sigma_max, uint64_t seed) : t_min(sigma_min), t_max(sigma_max), n(n_), root_seed(mix64(seed, 0x9E3779B97F4A7C15ULL)) {
        w_at_tmax     = std_default_randn(mix64(seed, 0xBF58476D1CE4E5B9ULL), n);
        const float f = std::sqrt(static_cast<float>(t_max - t_min));
        for (float& v : w_at_tmax) v *= f;
    }
    double clamp(double t) const { return std::min(std::max(t, t_min), t_max); }
    std::vector<float> bridge(double a, double c, const std::vector<float>& w_a, const std::vector<float>& w_c, double t, uint64_t node_seed, int depth) const {
        std::vector<float> r(n);
        if (depth <= 0 || c - a < 1e-9) {
            const float alpha = (c > a) ? static_cast<float>((t - a) / (c - a)) : 0.5f;
            const float om    = 1.0f - alpha;
            for (size_t k = 0; k < n; ++k) r[k] = om * w_a[k] + alpha * w_c[k];
            return r;
        }
        const double m = 0.5 * (a + c);
        const float sd = static_cast<float>(std::sqrt((c - m) * (m - a) / (c - a)));
        const std::vector<float> z = std_default_randn(node_seed, n);
        for (size_t k = 0; k < n; ++k) r[k] = 0.5f * (w_a[k] + w_c[k]) + sd * z[k];
        if (t == m) return r;
        if (t < m) return bridge(a, m, w_a, r, t, mix64(node_seed, 1), depth - 1);
        return bridge(m, c, r, w_c, t, mix64(node_seed, 2), depth - 1);
    }
    const std::vector<float>& w(double t) {
        auto it = cache.find(t);
        if (it != cache.end()) return it->second;
        const std::vector<float> zero(n, 0.0f);
        return cache.emplace(t, bridge(t_min, t_max, zero, w_at_tmax, t, root_seed, kMaxDepth)).first->second;
    }
    // unit-variance noise of the interval [sigma_a, sigma_b] into out
    void operator()(double sigma_a, double sigma_b, float* out) {
        const double a = clamp(std::min(sigma_a, sigma_b)), b = clamp(std::max(sigma_a, sigma_b));
        const std::vector<float> wb = w(b);  // (copy: the second lookup may rehash nothing — std::map keeps references valid — but keeps the reference's evaluation order w(b), w(a))
        const std::vector<float>& wa = w(a);
        const float span = static_cast<float>(std::max(std::abs(sigma_b - sigma_a), 1e-12));
        const float f    = 1.0f / std::sqrt(span);
        for (size_t k = 0; k < n; ++k) out[k] = (wb[k] - wa[k]) * f;
    }
};

// model(x, sigma, denoised, denoised_uncond) -> false on failure (denoised_uncond: nullptr unless the method steps along the unconditional prediction — the CFG++ methods);
// randn(out): one N(0,1) draw per element of x from the sampler's RNG stream(s) (sd::Tensor<float>::randn_like(x, rng));
// draw(b, count, out): `count` normals from image b's stream (rng->randn(count): the Brownian-tree seed of DPM++ 2M SDE BT); nb images of x.size() / nb floats each
template <class ModelFn, class RandnFn>
inline bool run_sampler_generic(int method, ModelFn&& model, std::vector<float>& x, const std::vector<float>& sigmas_in, RandnFn&& randn, float eta, bool flow, int nb = 1,
                                const std::function<void(int, uint32_t, float*)>& draw = nullptr) {
    const size_t n  = x.size();
    std::vector<float> sigmas = sigmas_in;
    const int steps = (int)sigmas.size() - 1;
    std::vector<float> den(n), x2(n), den2(n), d(n), nz;
    auto phi1_fn = [](float t) -> float {
        if (fabsf(t) < 1e-6f) return 1.0f + t * 0.5f + (t * t) / 6.0f;
        return (expf(t) - 1.0f) / t;
    };
    auto phi2_fn = [&](float t) -> float {
        if (fabsf(t) < 1e-6f) return 0.5f + t / 6.0f + (t * t) / 24.0f;
        return (phi1_fn(t) - 1.0f) / t;
    };
    auto t_fn     = [](float sigma) -> float { return (float)(-log((double)sigma)); };
    auto sigma_fn = [](float t) -> float { return (float)exp((double)-t); };
    switch (method) {
        case SM_HEUN:
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data(), nullptr, -(i + 1))) return false;
                const float dt = sigmas[i + 1] - sigmas[i];
                for (size_t k = 0; k < n; ++k) d[k] = (x[k] - den[k]) / sigmas[i];
                if (sigmas[i + 1] == 0) {
                    for (size_t k = 0; k < n; ++k) x[k] += d[k] * dt;
                } else {
                    for (size_t k = 0; k < n; ++k) x2[k] = x[k] + d[k] * dt;
                    if (!model(x2.data(), sigmas[i + 1], den2.data(), nullptr, i + 1)) return false;
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
                if (!model(x.data(), sigmas[i], den.data(), nullptr, -(i + 1))) return false;
                for (size_t k = 0; k < n; ++k) d[k] = (x[k] - den[k]) / sigmas[i];
                if (sigmas[i + 1] == 0) {
                    const float dt = sigmas[i + 1] - sigmas[i];
                    for (size_t k = 0; k < n; ++k) x[k] += d[k] * dt;
                } else {
                    const float sigma_mid = (float)exp(0.5f * (log((double)sigmas[i]) + log((double)sigmas[i + 1])));
                    const float dt_1 = sigma_mid - sigmas[i], dt_2 = sigmas[i + 1] - sigmas[i];
                    for (size_t k = 0; k < n; ++k) x2[k] = x[k] + d[k] * dt_1;
                    if (!model(x2.data(), sigma_mid, den2.data(), nullptr, i + 1)) return false;
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
                    if (!model(x.data(), sigmas[i], den.data(), nullptr, -(i + 1))) return false;
                    float sigma_down, sigma_up;
                    ancestral_step(sigmas[i], sigmas[i + 1], eta, sigma_down, sigma_up);
                    if (sigma_down == 0) {
                        x = den;
                    } else {
                        const float t = t_fn(sigmas[i]), t_next = t_fn(sigma_down), h = t_next - t, s = t + 0.5f * h, sigma_s = sigma_fn(s);
                        const float a1 = sigma_s / sigma_fn(t), b1 = (float)(exp((double)(-h * 0.5f)) - 1);
                        for (size_t k = 0; k < n; ++k) x2[k] = a1 * x[k] - b1 * den[k];
                        if (!model(x2.data(), sigma_s, den2.data(), nullptr, i + 1)) return false;
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
                if (!model(x.data(), sigma, den.data(), nullptr, (opt_first_step ? 1 : -1) * (i + 1))) return false;
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
                    if (!model(x2.data(), sigma_s, den2.data(), nullptr, i + 1)) return false;
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
                if (!model(x.data(), sigmas[i], den.data(), nullptr, i + 1)) return false;
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
                if (!model(x.data(), sigma, den.data(), nullptr, i + 1)) return false;
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
                if (!model(x.data(), sigmas[i], den.data(), nullptr, i + 1)) return false;
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
        case SM_DPMPP2M_SDE:      // sample_dpmpp_2m_sde, denoiser.hpp:1861-1907 (std::log / exp / expm1 / sqrt on floats: the float overloads)
        case SM_DPMPP2M_SDE_BT: {  // sample_dpmpp_2m_sde_bt, denoiser.hpp:1995-2057: the same update with Brownian-tree noise (one tree per image, seeded from its stream)
            std::vector<BrownianTree> trees;
            const size_t per = n / (size_t)(nb > 0 ? nb : 1);
            if (method == SM_DPMPP2M_SDE_BT) {
                double sigma_max = 0.0, sigma_min = std::numeric_limits<double>::infinity();
                for (float s : sigmas)
                    if (s > 0.0f) {
                        sigma_max = std::max(sigma_max, static_cast<double>(s));
                        sigma_min = std::min(sigma_min, static_cast<double>(s));
                    }
                if (sigma_max <= sigma_min) return true;  // (the reference returns x as it is)
                if (!draw) return false;
                for (int b = 0; b < nb; ++b) {
                    float two[2];
                    draw(b, 2, two);
                    uint64_t tree_seed = 0;
                    memcpy(&tree_seed, two, sizeof(tree_seed));
                    trees.emplace_back(per, sigma_min, sigma_max, tree_seed);
                }
            }
            std::vector<float> old;
            bool have_old = false;
            float h_last  = 0.f;
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data(), nullptr, i + 1)) return false;
                if (sigmas[i + 1] == 0.f) {
                    x = den;
                } else {
                    const float t = -logf(sigmas[i]), s = -logf(sigmas[i + 1]), h = s - t, eta_h = eta * h;
                    const float a = sigmas[i + 1] / sigmas[i] * expf(-eta_h);
                    const float b = -expm1f(-h - eta_h);
                    for (size_t k = 0; k < n; ++k) x[k] = a * x[k] + b * den[k];
                    if (have_old) {
                        const float r = h_last / h, c = 0.5f * b / r;
                        for (size_t k = 0; k < n; ++k) x[k] += c * (den[k] - old[k]);
                    }
                    if (eta > 0.f) {
                        nz.resize(n);
                        if (method == SM_DPMPP2M_SDE_BT)
                            for (int b2 = 0; b2 < nb; ++b2) trees[(size_t)b2]((double)sigmas[i], (double)sigmas[i + 1], nz.data() + (size_t)b2 * per);
                        else
                            randn(nz.data());
                        const float f = sigmas[i + 1] * sqrtf(-expm1f(-2.f * eta_h));
                        for (size_t k = 0; k < n; ++k) x[k] += nz[k] * f;
                    }
                    h_last = h;
                }
                old      = den;
                have_old = true;
            }
            return true;
        }
        case SM_RES_MULTISTEP: {  // sample_res_multistep, denoiser.hpp:2230-2306
            std::vector<float> old = x;
            bool have_old_sigma  = false;
            float old_sigma_down = 0.0f;
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data(), nullptr, i + 1)) return false;
                const float sigma_from = sigmas[i], sigma_to = sigmas[i + 1];
                float sigma_down, sigma_up, alpha_scale;
                ancestral_step3(sigma_from, sigma_to, eta, flow, sigma_down, sigma_up, alpha_scale);
                if (sigma_down == 0.0f || !have_old_sigma) {
                    const float dt = sigma_down - sigma_from;
                    for (size_t k = 0; k < n; ++k) x[k] += ((x[k] - den[k]) / sigma_from) * dt;
                } else {
                    const float t = -logf(sigma_from), t_old = -logf(old_sigma_down), t_next = -logf(sigma_down), t_prev = -logf(sigmas[i - 1]);
                    const float h = t_next - t, c2 = (t_prev - t_old) / h;
                    const float phi1_val = phi1_fn(-h), phi2_val = phi2_fn(-h);
                    float b1 = phi1_val - phi2_val / c2, b2 = phi2_val / c2;
                    if (!std::isfinite(b1)) b1 = 0.0f;
                    if (!std::isfinite(b2)) b2 = 0.0f;
                    const float sh = expf(-h);
                    for (size_t k = 0; k < n; ++k) {
                        const float in = b1 * den[k] + b2 * old[k];
                        x[k]           = sh * x[k] + h * in;
                    }
                }
                if (sigma_to > 0.0f && sigma_up > 0.0f) {
                    if (flow)
                        for (size_t k = 0; k < n; ++k) x[k] *= alpha_scale;
                    nz.resize(n);
                    randn(nz.data());
                    for (size_t k = 0; k < n; ++k) x[k] += nz[k] * sigma_up;
                }
                old            = den;
                old_sigma_down = sigma_down;
                have_old_sigma = true;
            }
            return true;
        }
        case SM_RES_2S: {  // sample_res_2s, denoiser.hpp:2308-2378
            const float c2 = 0.5f;
            std::vector<float> x0(n), eps1(n);
            for (int i = 0; i < steps; ++i) {
                const float sigma_from = sigmas[i], sigma_to = sigmas[i + 1];
                if (!model(x.data(), sigma_from, den.data(), nullptr, -(i + 1))) return false;
                float sigma_down, sigma_up, alpha_scale;
                ancestral_step3(sigma_from, sigma_to, eta, flow, sigma_down, sigma_up, alpha_scale);
                x0 = x;
                if (sigma_down == 0.0f || sigma_from == 0.0f) {
                    x = den;
                } else {
                    const float t = -logf(sigma_from), t_next = -logf(sigma_down), h = t_next - t;
                    const float a21 = c2 * phi1_fn(-h * c2), phi1_val = phi1_fn(-h), phi2_val = phi2_fn(-h);
                    const float b2 = phi2_val / c2, b1 = phi1_val - b2;
                    const float sigma_c2 = expf(-(t + h * c2)), ha = h * a21;
                    for (size_t k = 0; k < n; ++k) {
                        eps1[k] = den[k] - x0[k];
                        x2[k]   = x0[k] + eps1[k] * ha;
                    }
                    if (!model(x2.data(), sigma_c2, den2.data(), nullptr, i + 1)) return false;
                    for (size_t k = 0; k < n; ++k) {
                        const float eps2 = den2[k] - x0[k];
                        const float in   = b1 * eps1[k] + b2 * eps2;
                        x[k]             = x0[k] + h * in;
                    }
                }
                if (sigma_to > 0.0f && sigma_up > 0.0f) {
                    if (flow)
                        for (size_t k = 0; k < n; ++k) x[k] *= alpha_scale;
                    nz.resize(n);
                    randn(nz.data());
                    for (size_t k = 0; k < n; ++k) x[k] += nz[k] * sigma_up;
                }
            }
            return true;
        }
        case SM_ER_SDE: {  // sample_er_sde, denoiser.hpp:2380-2513
            constexpr int max_stage = 3, num_integration_points = 200;
            constexpr float num_integration_points_f = 200.0f;
            const float s_noise = eta;
            auto flow_sigma = [](float sigma) -> float {
                sigma = std::max(sigma, 1e-6f);
                sigma = std::min(sigma, 1.0f - 1e-4f);
                return sigma;
            };
            auto to_lambda = [&](float sigma) -> float {
                if (flow) {
                    sigma = flow_sigma(sigma);
                    return sigma / std::max(1.0f - sigma, 1e-6f);
                }
                return std::max(sigma, 1e-6f);
            };
            auto to_alpha = [&](float sigma) -> float {
                if (flow) {
                    sigma = flow_sigma(sigma);
                    return 1.0f - sigma;
                }
                return 1.0f;
            };
            auto noise_scaler = [](float v) -> float {
                v = std::max(v, 0.0f);
                return v * (expf(powf(v, 0.3f)) + 10.0f);
            };
            if (flow)
                for (size_t i = 0; i + 1 < sigmas.size(); ++i)
                    if (sigmas[i] > 1.0f) sigmas[i] = flow_sigma(sigmas[i]);
            std::vector<float> er_lambdas(sigmas.size(), 0.0f);
            for (size_t i = 0; i < sigmas.size(); ++i) er_lambdas[i] = to_lambda(sigmas[i]);
            std::vector<float> old = x, old_d = x, den_d(n);
            bool have_old = false, have_old_d = false;
            for (int i = 0; i < steps; ++i) {
                if (!model(x.data(), sigmas[i], den.data(), nullptr, i + 1)) return false;
                const int stage_used = std::min(max_stage, i + 1);
                if (sigmas[i + 1] == 0.0f) {
                    x = den;
                } else {
                    const float er_lambda_s = er_lambdas[i], er_lambda_t = er_lambdas[i + 1];
                    const float alpha_s = to_alpha(sigmas[i]), alpha_t = to_alpha(sigmas[i + 1]);
                    const float scaled_s = noise_scaler(er_lambda_s), scaled_t = noise_scaler(er_lambda_t);
                    const float r_alpha = alpha_s > 0.0f ? alpha_t / alpha_s : 0.0f;
                    const float r       = scaled_s > 0.0f ? scaled_t / scaled_s : 0.0f;
                    const float cx = r_alpha * r, cd = alpha_t * (1.0f - r);
                    for (size_t k = 0; k < n; ++k) x[k] = cx * x[k] + cd * den[k];
                    if (stage_used >= 2 && have_old) {
                        const float dt = er_lambda_t - er_lambda_s;
                        const float lambda_step_size = -dt / num_integration_points_f;
                        float s = 0.0f, s_u = 0.0f;
                        for (int p = 0; p < num_integration_points; ++p) {
                            const float lambda_pos = er_lambda_t + p * lambda_step_size;
                            const float scaled_pos = noise_scaler(lambda_pos);
                            if (scaled_pos <= 0.0f) continue;
                            s += 1.0f / scaled_pos;
                            if (stage_used >= 3 && have_old_d) s_u += (lambda_pos - er_lambda_s) / scaled_pos;
                        }
                        s *= lambda_step_size;
                        const float denom_d = er_lambda_s - er_lambdas[i - 1];
                        if (std::fabs(denom_d) > 1e-12f) {
                            const float coeff_d = alpha_t * (dt + s * scaled_t);
                            for (size_t k = 0; k < n; ++k) {
                                den_d[k] = (den[k] - old[k]) / denom_d;
                                x[k] += coeff_d * den_d[k];
                            }
                            if (stage_used >= 3 && have_old_d) {
                                const float denom_u = (er_lambda_s - er_lambdas[i - 2]) * 0.5f;
                                if (std::fabs(denom_u) > 1e-12f) {
                                    s_u *= lambda_step_size;
                                    const float coeff_u = alpha_t * (0.5f * dt * dt + s_u * scaled_t);
                                    for (size_t k = 0; k < n; ++k) {
                                        const float den_u = (den_d[k] - old_d[k]) / denom_u;
                                        x[k] += coeff_u * den_u;
                                    }
                                }
                            }
                            old_d      = den_d;
                            have_old_d = true;
                        }
                    }
                    const float noise_scale_sq = er_lambda_t * er_lambda_t - er_lambda_s * er_lambda_s * r * r;
                    if (s_noise > 0.0f && noise_scale_sq > 0.0f) {
                        const float noise_scale = alpha_t * std::sqrt(std::max(noise_scale_sq, 0.0f));
                        nz.resize(n);
                        randn(nz.data());
                        for (size_t k = 0; k < n; ++k) x[k] += nz[k] * noise_scale;
                    }
                }
                old      = den;
                have_old = true;
            }
            return true;
        }
        case SM_TCD: {  // sample_tcd, denoiser.hpp:2515-2579
            const float beta_start = 0.00085f, beta_end = 0.0120f;
            std::vector<double> alphas_cumprod(TIMESTEPS), compvis_sigmas(TIMESTEPS);
            for (int i = 0; i < TIMESTEPS; i++) {
                // (mixed types as the reference writes them: the float base is promoted by std::pow(float, int), the running product is double)
                alphas_cumprod[i] = (i == 0 ? 1.0f : alphas_cumprod[i - 1]) *
                                    (1.0f - std::pow(sqrtf(beta_start) + (sqrtf(beta_end) - sqrtf(beta_start)) * ((float)i / (TIMESTEPS - 1)), 2));
                compvis_sigmas[i] = std::sqrt((1 - alphas_cumprod[i]) / alphas_cumprod[i]);
            }
            auto get_timestep_from_sigma = [&](float s) -> int {
                auto it = std::lower_bound(compvis_sigmas.begin(), compvis_sigmas.end(), s);
                if (it == compvis_sigmas.begin()) return 0;
                if (it == compvis_sigmas.end()) return TIMESTEPS - 1;
                const int idx_high = static_cast<int>(std::distance(compvis_sigmas.begin(), it)), idx_low = idx_high - 1;
                if (std::abs(compvis_sigmas[idx_high] - s) < std::abs(compvis_sigmas[idx_low] - s)) return idx_high;
                return idx_low;
            };
            for (int i = 0; i < steps; ++i) {
                const float sigma_to    = sigmas[i + 1];
                const int prev_timestep = get_timestep_from_sigma(sigma_to);
                const int timestep_s    = (int)floor((1 - eta) * prev_timestep);
                const float sigma       = sigmas[i];
                if (!model(x.data(), sigma, den.data(), nullptr, i + 1)) return false;
                const float alpha_prod_t_prev = 1.0f / (sigma_to * sigma_to + 1.0f);
                const float alpha_prod_s      = static_cast<float>(alphas_cumprod[timestep_s]);
                const float beta_prod_s       = 1.0f - alpha_prod_s;
                const float c1 = std::sqrt(alpha_prod_s / alpha_prod_t_prev), c2 = std::sqrt(beta_prod_s / alpha_prod_t_prev);
                for (size_t k = 0; k < n; ++k) {
                    const float dk = (x[k] - den[k]) / sigma;
                    x[k]           = c1 * den[k] + c2 * dk;
                }
                if (eta > 0 && sigma_to > 0.0f) {
                    const float c3 = std::sqrt(alpha_prod_t_prev / alpha_prod_s), c4 = std::sqrt(1.0f / alpha_prod_t_prev - 1.0f / alpha_prod_s);
                    nz.resize(n);
                    randn(nz.data());
                    for (size_t k = 0; k < n; ++k) x[k] = c3 * x[k] + c4 * nz[k];
                }
            }
            return true;
        }
        case SM_LMS: {  // sample_lms, denoiser.hpp:2581-2685 with its defaults (no extra sample args): 1000 divisions, order 4, history shift 1
            const int divisions = 1000, shift = 1;
            const int max_order = std::min(4, steps);
            auto coeff = [&](const int order, const int m, const int j) -> float {
                const float a = sigmas[m], dx = (sigmas[m + 1] - a) / divisions, s = sigmas[m - j];
                const float b0 = a + 0.5f * dx;
                float sum = 0.0f;
                for (int h = 0; h < divisions; h++) {
                    const float b = h * dx + b0;
                    float prod    = 1.0f;
                    for (int k = 0; k < j; k++) {
                        const float t = sigmas[m - k];
                        prod *= (b - t) / (s - t);
                    }
                    for (int k = j + 1; k < order; k++) {
                        const float t = sigmas[m - k];
                        prod *= (b - t) / (s - t);
                    }
                    sum += prod;
                }
                return sum * dx;
            };
            std::vector<float> lms_coeff((size_t)std::max(max_order, 1));
            std::vector<std::vector<float>> hist;
            for (int i = 0; i < steps; ++i) {
                const float sigma = sigmas[i];
                if (!model(x.data(), sigma, den.data(), nullptr, i + 1)) return false;
                const int order = std::min(max_order, i + 1);
                for (int c = 0; c < order; c++) lms_coeff[(size_t)c] = coeff(order, i, c);
                std::vector<float> d_cur(n);
                for (size_t k = 0; k < n; ++k) {
                    d_cur[k] = (x[k] - den[k]) / sigma;
                    x[k] += d_cur[k] * lms_coeff[0];
                }
                if (max_order > 1) {
                    const int hist_size_p1 = (int)hist.size() + 1;
                    if (i) {
                        const int hist_max = (int)hist.size() - 1;
                        for (int c = 2; c <= order; c++) {
                            const std::vector<float>& hc = hist[(size_t)std::min(hist_max, hist_size_p1 - c + shift)];
                            const float lc               = lms_coeff[(size_t)c - 1];
                            for (size_t k = 0; k < n; ++k) x[k] += hc[k] * lc;
                        }
                    }
                    if (hist_size_p1 == max_order) hist.erase(hist.begin());
                    hist.push_back(std::move(d_cur));
                }
            }
            return true;
        }
        case SM_EULER_CFG_PP:      // sample_euler_cfg_pp, denoiser.hpp:2687-2705
        case SM_EULER_A_CFG_PP: {  // sample_euler_ancestral_cfg_pp, denoiser.hpp:2707-2733 (the two-value get_ancestral_step: no flow variant)
            std::vector<float> unc(n);
            for (int i = 0; i < steps; ++i) {
                const float sigma = sigmas[i];
                if (!model(x.data(), sigma, den.data(), unc.data(), i + 1)) return false;
                float to = sigmas[i + 1], sigma_up = 0.f;
                if (method == SM_EULER_A_CFG_PP) ancestral_step(sigmas[i], sigmas[i + 1], eta, to, sigma_up);
                for (size_t k = 0; k < n; ++k) {
                    const float dk = (x[k] - unc[k]) / sigma;
                    x[k]           = den[k] + dk * to;
                }
                if (method == SM_EULER_A_CFG_PP && sigmas[i + 1] > 0) {
                    nz.resize(n);
                    randn(nz.data());
                    for (size_t k = 0; k < n; ++k) x[k] += nz[k] * sigma_up;
                }
            }
            return true;
        }
        case SM_EULER_GE: {  // sample_gradient_estimation, denoiser.hpp:2736-2792 with gamma = 2 (no extra sample args)
            const float ge_gamma = 2.0f, omg = 1.0f - ge_gamma;
            std::vector<float> old_d;
            bool has_old_d = false;
            for (int i = 0; i < steps; ++i) {
                const float sigma = sigmas[i], sigma_to = sigmas[i + 1];
                if (!model(x.data(), sigma, den.data(), nullptr, i + 1)) return false;
                if (sigma_to == 0.f) {
                    x = den;
                } else {
                    float sigma_down, sigma_up, alpha_scale;
                    ancestral_step3(sigma, sigma_to, eta, flow, sigma_down, sigma_up, alpha_scale);
                    const float dt = sigma_down - sigma;
                    for (size_t k = 0; k < n; ++k) d[k] = (x[k] - den[k]) / sigma;
                    if (has_old_d) {
                        for (size_t k = 0; k < n; ++k) {
                            const float d_bar = d[k] * ge_gamma + old_d[k] * omg;
                            x[k] += d_bar * dt;
                        }
                    } else {
                        for (size_t k = 0; k < n; ++k) x[k] += d[k] * dt;
                    }
                    old_d     = d;
                    has_old_d = true;
                    if (sigma_up > 0.f) {
                        if (flow)
                            for (size_t k = 0; k < n; ++k) x[k] *= alpha_scale;
                        nz.resize(n);
                        randn(nz.data());
                        for (size_t k = 0; k < n; ++k) x[k] += nz[k] * sigma_up;
                    }
                }
            }
            return true;
        }
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

// AdaptiveProjectedGuidance::forward for the (cond, uncond) pair of ONE image of n floats — src/runtime/guidance.cpp:209-294: the guidance delta cond - uncond with momentum
// (the buffer is carried by the caller from step to step), rescaled to a norm threshold (norm measured at SDXL's standard resolution), its component parallel to cond
// scaled by eta; pred = cond + (scale - 1) * delta.  sd::Tensor<float>::sum accumulates in double and rounds once (tensor.hpp:340-347).  scale == 1: the reference returns cond.
struct ApgParams {
    float eta = 1.0f, momentum = 0.0f, norm_threshold = 0.0f, norm_threshold_smoothing = 0.0f;
    bool enabled() const { return eta != 1.0f || momentum != 0.0f || norm_threshold > 0.0f; }  // is_adaptive_projected_guidance_enabled, guidance.cpp:18-20
};
inline void apg_guided(const float* cond, const float* uncond, size_t n, float scale, const ApgParams& prm, std::vector<float>& momentum_buffer, float* out) {
    std::vector<float> deltas(n);
    for (size_t k = 0; k < n; ++k) deltas[k] = cond[k] - uncond[k];
    if (prm.momentum != 0.0f) {
        if (momentum_buffer.size() != n) momentum_buffer.assign(n, 0.0f);
        for (size_t k = 0; k < n; ++k) deltas[k] += prm.momentum * momentum_buffer[k];
        momentum_buffer = deltas;
    }
    auto sum_prod = [&](const float* a, const float* b) {
        double total = 0.0;
        for (size_t k = 0; k < n; ++k) {
            const float pr = a[k] * b[k];
            total += static_cast<double>(pr);
        }
        return static_cast<float>(total);
    };
    float diff_norm        = 0.0f;
    const int standard_res = 2 * 1024 / 8;
    if (prm.norm_threshold > 0.0f) diff_norm = std::sqrt(sum_prod(deltas.data(), deltas.data())) * standard_res / std::sqrt(static_cast<float>(n));
    float factor = 1.0f;
    if (prm.norm_threshold > 0.0f && diff_norm > 0.0f) {
        if (prm.norm_threshold_smoothing <= 0.0f) {
            factor = std::min(1.0f, prm.norm_threshold / diff_norm);
        } else {
            const float x = prm.norm_threshold / diff_norm;
            factor        = x / std::pow(1.0f + std::pow(x, 1.0f / prm.norm_threshold_smoothing), prm.norm_threshold_smoothing);
        }
    }
    for (size_t k = 0; k < n; ++k) deltas[k] *= factor;
    if (prm.eta != 1.0f) {
        const float cond_norm_sq = sum_prod(cond, cond);
        if (cond_norm_sq != 0.0f) {
            const float projection_scale = sum_prod(cond, deltas.data()) / cond_norm_sq;
            const float em1              = prm.eta - 1.0f;
            for (size_t k = 0; k < n; ++k) {
                const float pc = projection_scale * cond[k];
                deltas[k] += em1 * pc;
            }
        }
    }
    if (scale != 1.0f) {
        const float sm1 = scale - 1.0f;
        for (size_t k = 0; k < n; ++k) out[k] = cond[k] + sm1 * deltas[k];
    } else {
        for (size_t k = 0; k < n; ++k) out[k] = cond[k];
    }
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
