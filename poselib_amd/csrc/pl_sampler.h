// poselib_amd — counter-based minimal-sample generation.
//
// The reference sampler is splitmix64 with an `int` return (PoseLib/robust/sampling.cc:37-43);
// draw j (1-based) of a stream seeded with `seed` is  mix(seed + j*G), truncated to its low 32
// bits, reinterpreted as signed, sign-extended to 64 bits and reduced `% N` as unsigned
// (sampling.cc:50).  Because the generator is a pure function of the draw counter, a lane can
// produce the sample of any iteration from (seed, first-draw position of that iteration).
// Duplicate indices are redrawn (sampling.cc:52-58), which is why the position table exists.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
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

// PROSAC (sampling.cc:85-136): the subset-size recurrence is serial, so the samples are drawn on the HOST (one splitmix call per
// index) and handed to the generators explicitly.  Host-only.
struct ProsacSampler {
    uint64_t seed = 0, N = 0, max_it = 0;
    int K = 0;
    uint64_t pos = 0;        // splitmix draws consumed
    uint64_t sample_k = 1;   // sampling.cc:133
    uint64_t subset_sz = 0;  // sampling.cc:134
    std::vector<uint64_t> growth;

    void init(uint64_t seed_, uint64_t N_, int K_, uint64_t max_prosac_iterations) { // sampling.cc:105-135
        seed = seed_, N = N_, K = K_, max_it = max_prosac_iterations;
        growth.assign(std::max<uint64_t>(N, (uint64_t)K), 0);
        double T_n = (double)max_it;
        for (int i = 0; i < K; ++i)
            T_n *= static_cast<double>(K - i) / static_cast<double>(N - i);
        for (int n = 0; n < K; ++n)
            growth[n] = 1;
        uint64_t T_np = 1;
        for (uint64_t n = K; n < N; ++n) {
            const double T_n_next = T_n * (n + 1.0) / (n + 1.0 - K);
            T_np = (uint64_t)((double)T_np + std::ceil(T_n_next - T_n)); // `size_t += double` of sampling.cc:129
            growth[n] = T_np;
            T_n = T_n_next;
        }
        sample_k = 1;
        subset_sz = (uint64_t)K;
        pos = 0;
    }
    void draw(int count, uint64_t range, uint32_t *out) { // sampling.cc:46-61
        for (int i = 0; i < count; ++i) {
            for (;;) {
                const uint32_t v = (uint32_t)draw_at(seed, ++pos, range);
                bool fresh = true;
                for (int j = 0; j < i; ++j)
                    fresh = fresh && out[j] != v;
                if (fresh) {
                    out[i] = v;
                    break;
                }
            }
        }
    }
    void generate(uint32_t *sample) { // sampling.cc:85-103
        if (sample_k < max_it) {
            draw(K - 1, subset_sz - 1, sample);
            sample[K - 1] = (uint32_t)(subset_sz - 1);
            sample_k++;
            if (sample_k < max_it && sample_k > growth[subset_sz - 1])
                if (++subset_sz > N)
                    subset_sz = N;
        } else {
            draw(K, N, sample);
        }
    }
};

} // namespace pl
