// poselib_amd — counter-based minimal-sample generation.
//
// The reference sampler is splitmix64 with an `int` return (PoseLib/robust/sampling.cc:37-43);
// draw j (1-based) of a stream seeded with `seed` is  mix(seed + j*G), truncated to its low 32
// bits, reinterpreted as signed, sign-extended to 64 bits and reduced `% N` as unsigned
// (sampling.cc:50).  Because the generator is a pure function of the draw counter, a lane can
// produce the sample of any iteration from (seed, first-draw position of that iteration).
// Duplicate indices are redrawn (sampling.cc:52-58), which is why the position table exists.
#pragma once
#include "pl_math.h"

namespace pl {

constexpr uint64_t kSplitmixGamma = 0x9e3779b97f4a7c15ULL;

PL_HD uint64_t draw_at(uint64_t seed, uint64_t j1, uint64_t N) { // j1 = 1-based draw counter
    uint64_t z = seed + j1 * kSplitmixGamma;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z = z ^ (z >> 31);
    const int32_t r = (int32_t)(uint32_t)z;
    return (uint64_t)(int64_t)r % N;
}

// Draw K distinct indices starting after `pos` consumed draws.  Returns the number of draws
// consumed (K + rejections).
template <int K> PL_HD uint32_t draw_sample(uint64_t seed, uint64_t pos, uint64_t N, uint32_t *idx) {
    uint64_t j = pos;
    for (int i = 0; i < K; ++i) {
        for (;;) {
            const uint32_t v = (uint32_t)draw_at(seed, ++j, N);
            bool fresh = true;
            for (int k = 0; k < K; ++k)
                if (k < i && idx[k] == v)
                    fresh = false;
            if (fresh) {
                idx[i] = v;
                break;
            }
        }
    }
    return (uint32_t)(j - pos);
}

} // namespace pl
