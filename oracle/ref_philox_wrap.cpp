// TEST INFRASTRUCTURE (oracle/): C wrapper around the REFERENCE's own PhiloxRNG, compiled from the reference sources where they lie
// (/root/reference/src/core/rng_philox.hpp, rng.hpp — header-only, no ggml needed).  Built by oracle/Makefile into oracle/_ref/
// (git-ignored).  Used by tests/golden/make_philox_golden.py to generate the committed golden vectors and, when present, live by
// tests/test_host_logic.py.  Never linked into or loaded by the product.
#include <cstdint>
#include <cstring>
#include <vector>

#include "core/rng_philox.hpp"

extern "C" __attribute__((visibility("default"))) void ref_philox_randn(uint64_t seed, uint32_t n_calls_before, uint32_t n, float* out) {
    PhiloxRNG rng(seed);
    rng.manual_seed(seed);
    for (uint32_t i = 0; i < n_calls_before; ++i) (void)rng.randn(1);  // every call advances the offset by one (rng_philox.hpp:109)
    const std::vector<float> r = rng.randn(n);
    std::memcpy(out, r.data(), sizeof(float) * n);
}
