// rt_xorwow.h -- cuRAND-compatible XORWOW generator, host + gfx950 device.
//
// The reference draws every random number through cuRAND's default generator
// (curandState = XORWOW) and always seeds with subsequence 0, offset 0
// (src/main.cu:92,104; src/constant_medium.cuh:74), so no skip-ahead matrices
// are needed: seeding is ten integer operations and a draw is six.  rocRAND's
// xorwow uses different seed constants and a different uniform mapping
// (/opt/rocm/include/rocrand/rocrand_xorwow.h:113-116, rocrand_uniform.h:65-68),
// so it cannot be used here: every pixel would change.
//
// On the GPU the six state words live in VGPRs for the whole frame; the
// reference's 48-byte-per-pixel curandState array in HBM (main.cu:680,116,126)
// is never materialised.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

struct rt_xorwow {
    uint32_t v0, v1, v2, v3, v4, d;
};

// curand_init(seed, 0, 0, &state)
RT_HD void rt_xorwow_seed(rt_xorwow& s, uint64_t seed) {
    const uint32_t lo = (uint32_t)seed ^ 0xaad26b49u;
    const uint32_t hi = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t a = 1099087573u * lo;
    const uint32_t b = 2591861531u * hi;
    s.d = 6615241u + b + a;
    s.v0 = 123456789u + a;
    s.v1 = 362436069u ^ a;
    s.v2 = 521288629u + b;
    s.v3 = 88675123u ^ b;
    s.v4 = 5783321u + a;
}

// curand(&state)
RT_HD uint32_t rt_xorwow_next(rt_xorwow& s) {
    const uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}

// curand_uniform(&state): (0, 1].  Separate multiply and add (no FMA): the
// whole path is built with -ffp-contract=off.
RT_HD float rt_xorwow_uniform(rt_xorwow& s) {
    const uint32_t x = rt_xorwow_next(s);
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
