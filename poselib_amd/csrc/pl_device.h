// poselib_amd - small device-side helpers shared by the kernel translation units (kernels.hip, gen_rel.hip).
#pragma once
#include "pl_kernels.h"
#include "pl_sampler.h"

namespace pl {

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}


// Models per block of 1024 iterations (the first level of k_compact2's scan), accumulated by the generators themselves:
// one integer atomic per wavefront into a table the batch's control-block memset has zeroed.
// (n_nan: models with a NaN entry - statistics only, pl_ransac_stats.nan_hypotheses - go into a second table of the
// same shape; the atomics are spread over the blocks' table entries, one hot counter would serialise them)
__device__ __forceinline__ void count_models_of_wave(const GenerateArgs &g, uint32_t it, uint32_t n, uint32_t n_nan) {
    if (!g.blk_tot)
        return;
    const uint32_t s = wave_sum_u32(n);
    if ((threadIdx.x & 63) == 0 && s)
        atomicAdd(&g.blk_tot[it >> 10], s);
    if (g.blk_nan && __builtin_amdgcn_ballot_w64(n_nan != 0u)) {
        const uint32_t sn = wave_sum_u32(n_nan);
        if ((threadIdx.x & 63) == 0)
            atomicAdd(&g.blk_nan[it >> 10], sn);
    }
}
// NaN flag of a record that has been written (pl_math.h store_shadow)
__device__ __forceinline__ uint32_t record_is_nan(const double *rec) {
    return reinterpret_cast<const uint32_t *>(rec + kShadowOff)[13] != 0u ? 1u : 0u;
}


// the minimal sample of iteration `it`: explicit (PROSAC: drawn on the host) or regenerated from the draw position
template <int K> __device__ __forceinline__ void sample_of_iteration(const GenerateArgs &g, uint32_t it, uint32_t *idx) {
    if (g.samples) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            idx[k] = g.samples[(size_t)it * K + k];
    } else {
        draw_sample<K>(g.seed, g.pos_base + g.positions[it], g.pts.n, idx);
    }
}

} // namespace pl
