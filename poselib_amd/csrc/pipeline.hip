// poselib_amd — device-side bookkeeping kernels of one RANSAC batch (gfx950).  They remove the host from
// the per-batch critical path; everything here is integer / comparison work, deterministic, no atomics
// on the data path (one atomic counter orders nothing: the record list is re-sorted by hypothesis index).
//
//   k_sample_delta<K> + k_sample_orbit
//       Where does iteration i start in the splitmix64 draw stream?  Each iteration consumes K draws plus
//       one more per duplicate index (PoseLib/robust/sampling.cc:46-61), so the start positions form the
//       orbit  p -> p + delta(p)  of the function delta(p) = draws an iteration STARTING at p would consume.
//       delta is evaluated for every position in parallel; positions with delta != K are rare ("flags",
//       probability ~K^2/2N).  Between flags the orbit advances in strides of K, so one lane only has to hop
//       from flag to flag (~#duplicates hops per batch) and emit (iteration, position) segments; all lanes
//       then expand the segments into the per-iteration position table.
//   (k_count_blocks +) k_compact2   [the batched generators count per block themselves: GenerateArgs.blk_tot]
//       multi-workgroup exclusive scan of models-per-iteration -> hypothesis list in (iteration, model)
//       order and per-iteration hypothesis offsets.
//   k_finalize2 + k_records
//       chunk partials -> (count, score); then the running  best_minimal_inlier_count / _msac_score  scan of
//       PoseLib/robust/ransac_impl.h:113-123 over all hypotheses of the batch: a hypothesis is a "record"
//       iff count > max(previous counts) or score < min(previous scores).  Records (typically O(log H)) are
//       appended with their 128-byte model so the host fetches a few KB instead of every score.
#include "pl_kernels.h"
#include "pl_refine.h"
#include "pl_sampler.h"
#include <algorithm>
#include <cstdlib>

namespace pl {

__device__ __forceinline__ uint32_t wsum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}
// inclusive scan across the 64 lanes of a wavefront
__device__ __forceinline__ uint32_t wscan_add(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off, 64);
        if (lane >= off)
            v += u;
    }
    return v;
}
__device__ __forceinline__ uint32_t wscan_max(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off, 64);
        if (lane >= off)
            v = max(v, u);
    }
    return v;
}
__device__ __forceinline__ double wscan_min(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double u = __shfl_up(v, off, 64);
        if (lane >= off)
            v = fmin(v, u);
    }
    return v;
}

// ------------------------------------------------------------------------------------ sampler orbit
template <int K>
__device__ __forceinline__ void sample_delta_body(uint64_t seed, uint64_t pos_base, uint64_t N, uint32_t M,
                                                  uint8_t *delta, uint64_t *flagbits, uint32_t *zero,
                                                  uint32_t zero_words) {
    // the batch's control block and the generators' per-block model counts start at zero: the first kernel of the
    // batch clears them (everything that writes them is a later launch on the same stream) - no memset dispatch
    if (blockIdx.x == 0)
        for (uint32_t w = threadIdx.x; w < zero_words; w += 256)
            zero[w] = 0u;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    bool flag = false;
    if (p < M) {
        uint32_t idx[K];
        const uint32_t used = draw_sample<K>(seed, pos_base + p, N, idx);
        delta[p] = (uint8_t)min(used, 255u);
        flag = used != (uint32_t)K;
    }
    // bitmap of the positions whose iteration redraws: one 64-bit word per wavefront
    const unsigned long long m = __ballot(flag);
    if ((threadIdx.x & 63) == 0 && p < M)
        flagbits[p >> 6] = m;
}
template <int K>
__global__ __launch_bounds__(256) void k_sample_delta(uint64_t seed, uint64_t pos_base, uint64_t N, uint32_t M,
                                                      uint8_t *delta, uint64_t *flagbits, uint32_t *zero,
                                                      uint32_t zero_words) {
    sample_delta_body<K>(seed, pos_base, N, M, delta, flagbits, zero, zero_words);
}
// group form: problem = blockIdx.z, arguments from the group's table
template <int K> __global__ __launch_bounds__(256) void k_sample_delta_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    if (!g.active || blockIdx.x * 256u >= ((g.samp.M + 255u) & ~255u))
        return;
    sample_delta_body<K>(g.samp.seed, g.samp.pos_base, g.samp.N, g.samp.M, g.samp.delta, g.samp.flagbits,
                         reinterpret_cast<uint32_t *>(g.samp.ctl), g.samp.zero_words);
}

// Capacities of the orbit walk's LDS tables, two builds of the kernel:
//   large  8192 flagged positions (position + delta + successor), 4096 of them followed by pointer doubling, 4096 segments:
//          ~140 KB of LDS - the workgroup needs a CU to itself;
//   small  2048 / 2048 / 2048: 56 KB - fits next to two workgroups of the streaming scorers (50.7 KB each, three per CU),
//          i.e. it starts as soon as ONE scoring workgroup anywhere on the device retires.  With several groups in flight the
//          large build waited for a whole CU to drain, which under a scoring launch of another group means waiting for that
//          launch's tail (measured r03: 3.0 ms average in the grouped trace against 23 us alone).
// The host picks the small build when the expected number of flagged positions (M x P[a sample of K draws out of N repeats
// an index]) is at most kOrbitSmallExpected; more flags than the tables hold => orbit_error => the caller's fallback, as before.
constexpr int kMaxSegments = 4096;
constexpr int kMaxFlags = 8192;
constexpr int kParFlags = 4096;
constexpr int kSmallFlags = 2048;
constexpr double kOrbitSmallExpected = 1200.0; // (Poisson: 2048 is 24 standard deviations above)
constexpr int kOrbitWords = 4;   // bitmap words per lane and tile of phase 1

__device__ __forceinline__ uint32_t div_k(uint32_t x, int K) { // constant divisors compile to a multiply-high
    switch (K) {
    case 3:
        return x / 3u;
    case 4:
        return x >> 2;
    case 5:
        return x / 5u;
    default:
        return x / 7u;
    }
}

template <int MAXF, int PARF, int MAXSEG>
__device__ __forceinline__ void sample_orbit_body(const uint8_t *delta, const uint64_t *flagbits, uint32_t M, int K,
                                                  uint32_t B, uint64_t pos_base, uint32_t *positions, BatchCtl *ctl) {
    __shared__ uint32_t wave_tot[16], wave_off[16];
    __shared__ uint32_t flag_pos[MAXF];
    __shared__ uint8_t flag_delta[MAXF];
    __shared__ uint16_t flag_next[MAXF]; // next flag on the orbit that passes through this flag (0xffff: none)
    __shared__ uint32_t seg_iter[MAXSEG];
    __shared__ uint32_t seg_pos[MAXSEG];
    __shared__ uint32_t s_nseg, s_nflags, s_error, s_entry;
    // pointer doubling over the flags (phase 2b, parallel form)
    __shared__ uint32_t pj_w[PARF], pj_rank[PARF];
    __shared__ uint16_t pj_next[PARF], pj_hops[PARF];
    __shared__ uint8_t pj_mark[PARF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- phase 1: ordered list (in LDS) of the positions whose iteration would redraw (delta != K), from the
    // bitmap k_sample_delta wrote: tiles of 1024 x kOrbitWords words, workgroup scan of the per-lane counts,
    // ordered append (the delta byte is fetched for flagged positions only) ----
    if (threadIdx.x == 0) {
        s_nflags = 0;
        s_error = 0;
        s_entry = 0xffffffffu;
    }
    __syncthreads();
    const uint32_t nwords = (M + 63u) >> 6;
    for (uint32_t w0 = 0; w0 < nwords; w0 += 1024u * kOrbitWords) {
        const uint32_t wbase = w0 + threadIdx.x * kOrbitWords;
        uint64_t bits[kOrbitWords];
        uint32_t local = 0;
#pragma unroll
        for (int b = 0; b < kOrbitWords; ++b) {
            bits[b] = wbase + b < nwords ? flagbits[wbase + b] : 0ull;
            local += (uint32_t)__popcll(bits[b]);
        }
        const uint32_t inc = wscan_add(local, lane);
        if (lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t s = s_nflags;
            for (int w = 0; w < 16; ++w) {
                wave_off[w] = s;
                s += wave_tot[w];
            }
            s_nflags = s;
            if (s > (uint32_t)MAXF)
                s_error = 1u;
        }
        __syncthreads();
        if (!s_error && local) {
            uint32_t o = wave_off[wave] + inc - local;
#pragma unroll
            for (int b = 0; b < kOrbitWords; ++b) {
                uint64_t rest = bits[b];
                while (rest) {
                    const uint32_t q = ((wbase + b) << 6) + (uint32_t)__builtin_ctzll(rest);
                    rest &= rest - 1;
                    flag_pos[o] = q;
                    flag_delta[o] = delta[q];
                    ++o;
                }
            }
        }
        __syncthreads();
    }

    // ---- phase 2a: every flag looks up its successor on the orbit that passes through it: the first flag at
    // or after  pos + delta  on the same phase modulo K (flags are sorted; ~K candidates to inspect).  Flag -1 is the
    // virtual start (position 0, nothing consumed): its successor is the entry of the batch's orbit. ----
    const uint32_t F = s_error ? 0u : s_nflags;
    for (uint32_t j = threadIdx.x; j <= F; j += 1024u) {
        const bool start = j == F; // the virtual start flag
        const uint32_t from = start ? 0u : flag_pos[j] + (uint32_t)flag_delta[j];
        uint32_t k = start ? 0u : j + 1;
        uint32_t found = 0xffffu;
        for (; k < F; ++k) {
            const uint32_t q = flag_pos[k];
            if (q >= from && (q - from) - div_k(q - from, K) * (uint32_t)K == 0u) {
                found = k;
                break;
            }
        }
        if (start)
            s_entry = found;
        else
            flag_next[j] = (uint16_t)found;
    }
    __syncthreads();

    // ---- phase 2b: the orbit through the flags, as (iteration, position) segments.  A flag j visited by the orbit
    // starts a segment at iteration it_j = it_prev + (q_j - cur_prev) / K + 1 with position cur_j = q_j + delta_j.
    // Up to PARF flags the successor links are followed by POINTER DOUBLING: every flag carries its 2^k-th successor
    // and the iterations that lie between; the flags reachable from the entry are marked round by round (after round k
    // all flags within 2^(k+1) hops) and receive their iteration index and hop count on the way - 12 rounds of all
    // lanes instead of one lane hopping ~300 times through LDS (22 us -> 4 us at config 1).  Beyond that: one lane. ----
    if (!s_error && F <= (uint32_t)PARF) {
        constexpr int kPer = PARF / 1024;
        const uint32_t entry = s_entry;
        for (uint32_t j = threadIdx.x; j < F; j += 1024u) {
            const uint32_t nj = flag_next[j];
            pj_next[j] = (uint16_t)nj;
            pj_w[j] = nj != 0xffffu ? div_k(flag_pos[nj] - (flag_pos[j] + (uint32_t)flag_delta[j]), K) + 1u : 0u;
            pj_mark[j] = j == entry ? 1 : 0;
            pj_rank[j] = j == entry ? div_k(flag_pos[j], K) + 1u : 0u;
            pj_hops[j] = 0;
        }
        if (threadIdx.x == 0) {
            s_nseg = 1;
            seg_iter[0] = 0;
            seg_pos[0] = 0;
        }
        __syncthreads();
        for (uint32_t step = 1; step < F; step <<= 1) { // (wave-uniform trip count)
            uint32_t nj[kPer], wj[kPer], nn[kPer], wn[kPer], rk[kPer], hp[kPer];
            bool mk[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const uint32_t j = threadIdx.x + 1024u * u;
                nj[u] = 0xffffu;
                if (j < F) {
                    nj[u] = pj_next[j], wj[u] = pj_w[j], mk[u] = pj_mark[j] != 0, rk[u] = pj_rank[j], hp[u] = pj_hops[j];
                    if (nj[u] != 0xffffu)
                        nn[u] = pj_next[nj[u]], wn[u] = pj_w[nj[u]];
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const uint32_t j = threadIdx.x + 1024u * u;
                if (j < F && nj[u] != 0xffffu) {
                    if (mk[u]) { // (the marked flags lie on one path: their successors are distinct)
                        pj_mark[nj[u]] = 1;
                        pj_rank[nj[u]] = rk[u] + wj[u];
                        pj_hops[nj[u]] = (uint16_t)(hp[u] + step);
                    }
                    pj_next[j] = (uint16_t)nn[u];
                    pj_w[j] = wj[u] + wn[u];
                }
            }
            __syncthreads();
        }
        // segments of the visited flags whose iteration still belongs to the batch (the serial walk stops at the first
        // one that does not); their hop count is their place in the table
        for (uint32_t j = threadIdx.x; j < F; j += 1024u) {
            if (!pj_mark[j] || pj_rank[j] > B)
                continue;
            const uint32_t idx = (uint32_t)pj_hops[j] + 1u;
            if (flag_delta[j] == 255u || idx >= (uint32_t)MAXSEG) {
                s_error = 1;
                continue;
            }
            seg_iter[idx] = pj_rank[j];
            seg_pos[idx] = flag_pos[j] + (uint32_t)flag_delta[j];
            atomicMax(&s_nseg, idx + 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t err = s_error;
            const uint32_t s = s_nseg - 1;
            const uint64_t end = (uint64_t)seg_pos[s] + (uint64_t)(B - seg_iter[s]) * (uint64_t)K;
            if (end + 255 > M)
                err = 1; // the window of evaluated positions was too small: the host retries / falls back
            ctl->pos_after = pos_base + end;
            ctl->orbit_error = err;
            s_error = err;
        }
    } else if (threadIdx.x == 0) {
        uint32_t nseg = 1, err = s_error;
        seg_iter[0] = 0;
        seg_pos[0] = 0;
        uint32_t cur = 0, it = 0, j = s_entry;
        while (!err && j != 0xffffu) {
            const uint32_t q = flag_pos[j];
            const uint32_t before = div_k(q - cur, K);
            if (it + before >= B)
                break; // the batch ends before that iteration
            const uint32_t d = flag_delta[j];
            if (d == 255u || nseg >= (uint32_t)MAXSEG) {
                err = 1;
                break;
            }
            it += before + 1;
            cur = q + d;
            seg_iter[nseg] = it;
            seg_pos[nseg] = cur;
            ++nseg;
            j = flag_next[j];
        }
        // position of iteration B == draws consumed by the batch, from the last segment of the table
        const uint32_t s = nseg - 1;
        const uint64_t end = (uint64_t)seg_pos[s] + (uint64_t)(B - seg_iter[s]) * (uint64_t)K;
        if (end + 255 > M)
            err = 1; // the window of evaluated positions was too small: the host retries / falls back
        ctl->pos_after = pos_base + end;
        ctl->orbit_error = err;
        s_nseg = nseg;
        s_error = err;
    }
    __syncthreads();

    // ---- phase 3: expand segments to per-iteration positions (relative to pos_base).  Every wavefront takes a
    // contiguous sixteenth of the iterations, its lanes stride through it by 64: a lane's iterations ascend in small
    // steps, so its segment index only creeps forward - one compare per iteration against the start of the next
    // segment (kept in a register), a short linear probe when it is crossed, a binary search if the probe fails ----
    const uint32_t nseg = s_nseg;
    const uint32_t per_wave = ((B + 15u) / 16u + 63u) & ~63u;
    const uint32_t i0 = (uint32_t)wave * per_wave, i1 = min(B, i0 + per_wave);
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) { // last segment with seg_iter <= i0 (wave-uniform)
        const uint32_t mid = (lo + hi) >> 1;
        if (seg_iter[mid] <= i0)
            lo = mid;
        else
            hi = mid;
    }
    uint32_t cur_it = seg_iter[lo], cur_pos = seg_pos[lo];
    uint32_t next_it = lo + 1 < nseg ? seg_iter[lo + 1] : 0xffffffffu;
    for (uint32_t i = i0 + (uint32_t)lane; i < i1; i += 64) {
        if (i >= next_it) {
            int probe = 0;
            do {
                ++lo;
                next_it = lo + 1 < nseg ? seg_iter[lo + 1] : 0xffffffffu;
            } while (i >= next_it && ++probe < 4);
            if (i >= next_it) {
                hi = nseg;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (seg_iter[mid] <= i)
                        lo = mid;
                    else
                        hi = mid;
                }
                next_it = lo + 1 < nseg ? seg_iter[lo + 1] : 0xffffffffu;
            }
            cur_it = seg_iter[lo];
            cur_pos = seg_pos[lo];
        }
        positions[i] = cur_pos + (i - cur_it) * (uint32_t)K;
    }
}
template <bool SMALL>
__global__ __launch_bounds__(1024) void k_sample_orbit(const uint8_t *delta, const uint64_t *flagbits, uint32_t M, int K,
                                                       uint32_t B, uint64_t pos_base, uint32_t *positions,
                                                       BatchCtl *ctl) {
    if constexpr (SMALL)
        sample_orbit_body<kSmallFlags, kSmallFlags, kSmallFlags>(delta, flagbits, M, K, B, pos_base, positions, ctl);
    else
        sample_orbit_body<kMaxFlags, kParFlags, kMaxSegments>(delta, flagbits, M, K, B, pos_base, positions, ctl);
}
template <bool SMALL> __global__ __launch_bounds__(1024) void k_sample_orbit_g(const GroupArgs *ga, int K) {
    const GroupArgs &g = ga[blockIdx.z];
    if (!g.active)
        return;
    if constexpr (SMALL)
        sample_orbit_body<kSmallFlags, kSmallFlags, kSmallFlags>(g.samp.delta, g.samp.flagbits, g.samp.M, K, g.samp.B, g.samp.pos_base,
                                                               g.samp.positions, g.samp.ctl);
    else
        sample_orbit_body<kMaxFlags, kParFlags, kMaxSegments>(g.samp.delta, g.samp.flagbits, g.samp.M, K, g.samp.B, g.samp.pos_base,
                                                             g.samp.positions, g.samp.ctl);
}
// expected number of flagged positions among M: a position is flagged when the K draws starting there repeat an index
static bool orbit_small_build(uint32_t M, uint64_t N, int K) {
    static const bool off = std::getenv("POSELIB_AMD_ORBIT_LARGE") != nullptr;
    if (off || N == 0)
        return false;
    double distinct = 1.0;
    for (int i = 1; i < K; ++i)
        distinct *= 1.0 - std::min(1.0, (double)i / (double)N);
    return (double)M * (1.0 - distinct) <= kOrbitSmallExpected;
}

// ------------------------------------------------------------------------------------ compaction
__global__ __launch_bounds__(1024) void k_count_blocks(const uint32_t *num_models, uint32_t B, uint32_t *blk_tot) {
    __shared__ uint32_t wt[16];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = (i < B) ? num_models[i] : 0u;
    const uint32_t s = wsum_u32(v);
    if ((threadIdx.x & 63) == 0)
        wt[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; ++w)
            t += wt[w];
        blk_tot[blockIdx.x] = t;
    }
}

__device__ __forceinline__ void compact2_body(const uint32_t *num_models, uint32_t B, int maxm,
                                              const uint32_t *blk_tot, uint32_t *slots, uint32_t *offsets,
                                              BatchCtl *ctl, uint32_t nblocks, uint32_t *host_offsets) {
    __shared__ uint32_t wt[16], wo[16], wb[16], wn[16];
    __shared__ uint32_t s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // models of the blocks before this one (and, in the last block, the generators' NaN-model counts: second table behind
    // blk_tot, statistics): all lanes fetch, one wave sum each - one lane walking the table paid ~1 us per entry
    uint32_t before = 0, nan_part = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 1024u)
        before += blk_tot[j];
    const bool last = blockIdx.x == nblocks - 1;
    if (last)
        for (uint32_t j = threadIdx.x; j < nblocks; j += 1024u)
            nan_part += blk_tot[nblocks + j];
    before = wsum_u32(before);
    nan_part = wsum_u32(nan_part);
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t nm = (i < B) ? num_models[i] : 0u;
    const uint32_t inc = wscan_add(nm, lane);
    if (lane == 63)
        wt[wave] = inc;
    if (lane == 0) {
        wb[wave] = before;
        wn[wave] = nan_part;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0, b = 0, nan = 0;
        for (int w = 0; w < 16; ++w) {
            wo[w] = s;
            s += wt[w];
            b += wb[w];
            nan += wn[w];
        }
        s_base = b;
        if (last) {
            ctl->num_hyp = b + s;
            ctl->nan_hyp = nan;
        }
    }
    __syncthreads();
    if (i < B) {
        const uint32_t o = s_base + wo[wave] + inc - nm;
        offsets[i] = o;
        if (host_offsets)
            host_offsets[i] = o;
        for (uint32_t m = 0; m < nm; ++m) {
            const uint32_t slot = i * (uint32_t)maxm + m;
            slots[o + m] = slot;
        }
    }
}

__global__ __launch_bounds__(1024) void k_compact2(const uint32_t *num_models, uint32_t B, int maxm,
                                                   const uint32_t *blk_tot, uint32_t *slots, uint32_t *offsets,
                                                   const double *models, float *shadow_compact, double *compact64,
                                                   BatchCtl *ctl) {
    compact2_body(num_models, B, maxm, blk_tot, slots, offsets, ctl, gridDim.x, nullptr);
}
__global__ __launch_bounds__(1024) void k_compact2_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    const uint32_t nb = (g.comp.B + 1023u) / 1024u;
    if (!g.active || blockIdx.x >= nb)
        return;
    compact2_body(g.comp.num_models, g.comp.B, g.comp.maxm, g.comp.blk_tot, g.comp.slots, g.comp.offsets, g.comp.ctl, nb,
                  g.comp.host_offsets);
}

// Hypothesis-ordered copies of the records for the streaming scorer: 12 lanes move the 192 bytes of one record
// (16 B each, contiguous on both sides): bytes 0..127 = the fp64 model -> compact64, 128..191 = the fp32 shadow.
__device__ __forceinline__ void gather_one(BatchCtl *ctl, const uint32_t *slots, const double *models,
                                           float *shadow_compact, double *compact64, uint64_t t) {
    static_assert(kModelStride == 24 && kModelDoubles == 16 && kShadowOff == 16, "record layout");
    const uint32_t H = ctl->num_hyp;
    const uint32_t k = (uint32_t)(t / 12), part = (uint32_t)(t % 12);
    bool nan_model = false;
    if (k < H) {
        const uint4 v = reinterpret_cast<const uint4 *>(models + (size_t)slots[k] * kModelStride)[part];
        if (part < 8)
            reinterpret_cast<uint4 *>(compact64 + (size_t)k * kModelDoubles)[part] = v;
        else
            reinterpret_cast<uint4 *>(shadow_compact + (size_t)k * 16)[part - 8] = v;
    }
    (void)nan_model;
}
__global__ __launch_bounds__(256) void k_gather_models(BatchCtl *ctl, const uint32_t *slots,
                                                       const double *models, float *shadow_compact, double *compact64) {
    gather_one(ctl, slots, models, shadow_compact, compact64, (uint64_t)blockIdx.x * 256 + threadIdx.x);
}

// ------------------------------------------------------------------------------------ finalize + records
constexpr int kRecBlocks = 256; // hypothesis chunks (contiguous), shared by k_finalize2 and k_records

__device__ __forceinline__ void finalize2_body(const FinalizeArgs &f, uint32_t *blk_max, double *blk_min) {
    __shared__ uint32_t wmax[4];
    __shared__ double wmin[4];
    const uint32_t H = *f.num_hyp;
    const uint32_t per = (H + gridDim.x - 1) / gridDim.x;
    const uint32_t k0 = min(H, blockIdx.x * per), k1 = min(H, k0 + per);
    uint32_t bmax = 0;
    double bmin = 1.7976931348623157e308;
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) {
        uint32_t c = 0;
        double s = 0.0;
        for (uint32_t ch = 0; ch < f.chunks; ++ch) {
            c += f.part_count[(size_t)ch * f.hyp_capacity + k];
            s += f.part_score[(size_t)ch * f.hyp_capacity + k];
        }
        const double sc = s + (double)(f.n_points - c) * f.thr2; // utils.cc:63 / :193-197
        f.count[k] = c;
        f.score[k] = sc;
        bmax = max(bmax, c);
        bmin = fmin(bmin, sc);
    }
    if (blk_max) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            bmax = max(bmax, (uint32_t)__shfl_xor(bmax, off, 64));
            bmin = fmin(bmin, __shfl_xor(bmin, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            wmax[threadIdx.x >> 6] = bmax;
            wmin[threadIdx.x >> 6] = bmin;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) {
                bmax = max(bmax, wmax[w]);
                bmin = fmin(bmin, wmin[w]);
            }
            blk_max[blockIdx.x] = bmax;
            blk_min[blockIdx.x] = bmin;
        }
    }
}
__global__ __launch_bounds__(256) void k_finalize2(FinalizeArgs f, uint32_t *blk_max, double *blk_min) {
    finalize2_body(f, blk_max, blk_min);
}
__global__ __launch_bounds__(256) void k_finalize2_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    if (!g.active)
        return;
    finalize2_body(g.rec.f, g.rec.blk_max, g.rec.blk_min);
}

__device__ __forceinline__ void records_body(const uint32_t *num_hyp, const uint32_t *count, const double *score,
                                             const uint32_t *slots, const double *models, const uint32_t *blk_max,
                                             const double *blk_min, uint32_t init_max, double init_min,
                                             RecordMeta *rec_meta, double *rec_models, uint32_t rec_cap,
                                             BatchCtl *ctl, RecordMeta *host_meta, double *host_models,
                                             uint32_t host_cap) {
    __shared__ uint32_t wmax[4];
    __shared__ double wmin[4];
    __shared__ uint32_t s_runmax;
    __shared__ double s_runmin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t H = *num_hyp;
    const uint32_t per = (H + gridDim.x - 1) / gridDim.x;
    const uint32_t k0 = min(H, blockIdx.x * per), k1 = min(H, k0 + per);
    { // state of the sequential loop when it reaches this chunk: max / min over the chunks in front of it
        static_assert(kRecBlocks == 256, "one earlier chunk per thread");
        uint32_t m = (threadIdx.x < blockIdx.x) ? blk_max[threadIdx.x] : 0u;
        double s = (threadIdx.x < blockIdx.x) ? blk_min[threadIdx.x] : 1.7976931348623157e308;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
            s = fmin(s, __shfl_xor(s, off, 64));
        }
        if (lane == 0) {
            wmax[wave] = m;
            wmin[wave] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t mm = init_max;
            double ss = init_min;
            for (int w = 0; w < 4; ++w) {
                mm = max(mm, wmax[w]);
                ss = fmin(ss, wmin[w]);
            }
            s_runmax = mm;
            s_runmin = ss;
        }
        __syncthreads();
    }
    for (uint32_t t0 = k0; t0 < k1; t0 += 256) {
        const uint32_t k = t0 + threadIdx.x;
        const bool live = k < k1;
        const uint32_t c = live ? count[k] : 0u;
        const double s = live ? score[k] : 1.7976931348623157e308;
        const uint32_t imax = wscan_max(c, lane);
        const double imin = wscan_min(s, lane);
        if (lane == 63) {
            wmax[wave] = imax;
            wmin[wave] = imin;
        }
        __syncthreads();
        // exclusive running values in front of this lane
        uint32_t emax = s_runmax;
        double emin = s_runmin;
        for (int w = 0; w < wave; ++w) {
            emax = max(emax, wmax[w]);
            emin = fmin(emin, wmin[w]);
        }
        const uint32_t pmax = __shfl_up(imax, 1, 64);
        const double pmin = __shfl_up(imin, 1, 64);
        if (lane > 0) {
            emax = max(emax, pmax);
            emin = fmin(emin, pmin);
        }
        // ransac_impl.h:114-116 with a margin on the score: these sums are in tree order, the decision is taken on the
        // sequentially summed scores k_score_seq computes for the listed candidates (the host applies the exact rule)
        if (live && (c > emax || s < emin * (1.0 + 1e-9))) {
            const uint32_t r = atomicAdd(&ctl->num_records, 1u);
            if (r < rec_cap) {
                const uint32_t slot = slots ? slots[k] : k;
                RecordMeta m;
                m.k = k;
                m.slot = slot;
                m.count = c;
                m.pad = 0;
                m.score = s;
                rec_meta[r] = m;
                for (int i = 0; i < kModelStride; ++i)
                    rec_models[(size_t)r * kModelStride + i] = models[(size_t)slot * kModelStride + i];
                if (r < host_cap) { // the first models go straight to pinned host memory (no copy dispatch); their
                                    // (count, score) follow from k_score_seq
                    for (int i = 0; i < kModelStride; ++i)
                        host_models[(size_t)r * kModelStride + i] = models[(size_t)slot * kModelStride + i];
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 255) { // carry the running state to the next tile
            s_runmax = max(emax, c);
            s_runmin = fmin(emin, s);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_records(const uint32_t *num_hyp, const uint32_t *count, const double *score,
                                                 const uint32_t *slots, const double *models, const uint32_t *blk_max,
                                                 const double *blk_min, uint32_t init_max, double init_min,
                                                 RecordMeta *rec_meta, double *rec_models, uint32_t rec_cap,
                                                 BatchCtl *ctl, RecordMeta *host_meta, double *host_models,
                                                 uint32_t host_cap) {
    records_body(num_hyp, count, score, slots, models, blk_max, blk_min, init_max, init_min, rec_meta, rec_models, rec_cap,
                 ctl, host_meta, host_models, host_cap);
}
__global__ __launch_bounds__(256) void k_records_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    if (!g.active)
        return;
    const RecordsArgs &r = g.rec;
    records_body(r.f.num_hyp, r.f.count, r.f.score, r.slots, r.models, r.blk_max, r.blk_min, r.init_max, r.init_min,
                 r.rec_meta, r.rec_models, r.rec_cap, r.ctl, r.host_meta, r.host_models, r.host_meta ? r.host_cap : 0u);
}

// ------------------------------------------------------------------------------------ fp16 hypothesis operands
// A-operand rows of v_mfma_f32_32x32x16_f16 for k_score_mfma (kernels.hip; bounds in pl_prefilter.h).  An inlier's residual
// vector (z_0 - x z_2, z_1 - y z_2) is shorter than thr z_2, so its component along ANY unit direction d = (c, s) is: three
// directions 120 degrees apart bound the inlier disc by a triangle, and each half-plane is LINEAR in the sixteen
// per-correspondence numbers  (X, X_lo, 1, w | p X, (p X)_lo, p, |p|),  p = c x + s y:
//     F  =  (thr R_2 - (c R_0 + s R_1)) . X  +  (thr t_2 - (c t_0 + s t_1) + g)  +  w  +  p (R_2 . X)  +  p t_2  +  Tm |p|
// so the matrix pipe delivers the three signed distances of a pair directly, slack included, and the vector ALU only ORs
// three sign bits (one v_or3).  One instruction = 32 hypotheses x 32 correspondences of ONE direction; the three
// instructions share the first k block of the correspondence side and the second k block of the hypothesis side.  Stored per
// hypothesis as four rows of 16 B:  [block 0 of direction 0][of direction 1][of direction 2][block 1 (all three)]
//     block 0 = (c_0, c_1, c_2, c_0, c_1, c_2, const, 1),  c = thr R_2 - (c R_0 + s R_1),  const rounded UP
//     block 1 = (R_20, R_21, R_22, R_20, R_21, R_22, t_2, Tm)
// g = 2^-16 max|t_c| + c16 is the hypothesis' share of the slack (+inf: evaluate every point exactly, -inf: NaN model, no
// inliers; the other entries are zero then), Tm = 2^-10 (1 + 2^-6) max|t_c| the factor of |p| (pl_prefilter.h).
// shadow_of(k): the fp32 shadow (16 floats) of hypothesis k < H
template <typename ShadowOf>
__device__ __forceinline__ void shadow16_one(uint32_t k, uint32_t H, ShadowOf shadow_of, float c16, float thr,
                                             uint4 *__restrict__ out) {
    if (k >= ((H + 31u) & ~31u))
        return; // groups of 32 past the last hypothesis are never read
    Abs16Model m;
    pf16_abs_model(k < H ? shadow_of(k) : nullptr, c16, thr, m); // (pl_prefilter.h: the host test build runs the same)
    uint4 *dst = out + (size_t)k * 4;
    auto row = [](const uint16_t *h) {
        return make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                          (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
    };
#pragma unroll
    for (int d = 0; d < kAbs16Dirs; ++d)
        dst[d] = row(m.d[d]);
    dst[3] = row(m.b1);
}

__global__ __launch_bounds__(256) void k_shadow16(const uint32_t *num_hyp, const float *__restrict__ shadow,
                                                  uint32_t capacity16, float g16, float c16, float thr,
                                                  uint4 *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= capacity16)
        return;
    (void)g16;
    shadow16_one(k, *num_hyp, [&](uint32_t kk) { return shadow + (size_t)kk * 16; }, c16, thr, out);
}

// k_gather_models and k_shadow16 as ONE launch (both only depend on k_compact2's hypothesis list): blocks below
// `gather_blocks` copy the records, the blocks above build the fp16 operand blocks straight from the records' shadows.
__global__ __launch_bounds__(256) void k_gather_shadow16(BatchCtl *ctl, const uint32_t *slots, const double *models,
                                                         float *shadow_compact, double *compact64,
                                                         uint32_t gather_blocks, uint32_t capacity16, float g16, float c16,
                                                         float thr, uint4 *__restrict__ out16) {
    if (blockIdx.x < gather_blocks) {
        gather_one(ctl, slots, models, shadow_compact, compact64, (uint64_t)blockIdx.x * 256 + threadIdx.x);
        return;
    }
    const uint32_t k = (blockIdx.x - gather_blocks) * 256 + threadIdx.x;
    if (k >= capacity16)
        return;
    shadow16_one(k, ctl->num_hyp,
                 [&](uint32_t kk) { return reinterpret_cast<const float *>(models + (size_t)slots[kk] * kModelStride + kShadowOff); },
                 c16, thr, out16);
}

// Operands of k_score_mfma2 (Sampson scores on the matrix cores): 96 B per hypothesis = six 16-byte k blocks
// [C~ k0-7][k8-15][k16-23][k24-31][S~ k0-7][k8-15], built from the fp64 record (pl_prefilter.h pf16_sampson_model).
__device__ __forceinline__ void sampson16_one(uint32_t k, uint32_t H, uint32_t cap, const uint32_t *__restrict__ slots,
                                              const double *__restrict__ models, uint4 *__restrict__ out) {
    if (k >= cap)
        return;
    Sampson16Operand o;
    if (k < H) {
        const double *rec = models + (size_t)slots[k] * kModelStride;
        const bool nan_model = reinterpret_cast<const uint32_t *>(rec + kShadowOff)[13] != 0u;
        pf16_sampson_model(rec + kMatOff, nan_model, o);
    } else { // not a hypothesis: never a candidate
        const double zero[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        pf16_sampson_model(zero, true, o);
    }
    uint4 *dst = out + (size_t)k * 6;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const uint16_t *h = b < 4 ? o.c + 8 * b : o.s + 8 * (b - 4);
        dst[b] = make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                            (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
    }
}
__global__ __launch_bounds__(256) void k_sampson16(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t cap,
                                                   uint4 *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint32_t H = ctl->num_hyp;
    // (rows past the last hypothesis of the last unit: one pad of kSampson16Pad rows is all a partial group can reach)
    if (k >= min(cap, (H + (uint32_t)kSampson16Pad)))
        return;
    sampson16_one(k, H, cap, slots, models, out);
}

// Operands of k_score_mfmah (homography on the matrix cores): four rows (V_0, V_1, U, S) of 32 k slots per hypothesis,
// built from the fp64 record (pl_prefilter.h pf16_hom_model).  Stored per group of 8 hypotheses as four blocks of 32 rows x
// 16 B: [k slots 0..7][8..15][16..23][24..31], row 4 j + r for hypothesis j of the group.
__device__ __forceinline__ void hom16_one(uint32_t k, uint32_t H, const uint32_t *__restrict__ slots,
                                          const double *__restrict__ models, float thr, uint4 *__restrict__ out) {
    if (k >= ((H + 7u) & ~7u))
        return; // groups past the last hypothesis are never read
    Hom16Model o;
    if (k < H) {
        const double *rec = models + (size_t)slots[k] * kModelStride;
        const bool nan_model = reinterpret_cast<const uint32_t *>(rec + kShadowOff)[13] != 0u;
        pf16_hom_model(rec + kMatOff, nan_model, thr, o);
    } else { // not a hypothesis: never a candidate
        const double zero[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        pf16_hom_model(zero, true, thr, o);
    }
    uint4 *grp = out + (size_t)(k >> 3) * 128;
    const uint32_t j = k & 7u;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint16_t *h = o.r[r] + 8 * b;
            grp[32 * b + 4 * j + r] = make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                                                 (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
        }
}
__global__ __launch_bounds__(256) void k_hom16(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t cap,
                                               float thr, uint4 *__restrict__ out) {
    const uint32_t H = ctl->num_hyp;
    const uint32_t end = min(cap, (H + 7u) & ~7u);
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < end; k += gridDim.x * 256)
        hom16_one(k, H, slots, models, thr, out);
}

// Group form.  The grid is NOT sized by the record capacity (B x slots per iteration: 1.6 M slots for a 100 k-iteration
// 5-point batch of which ~60 k hold a hypothesis - a capacity-sized grid was ~80 k empty workgroups per problem, whose
// dispatch alone cost 8 % of the grouped 5-point step): a bounded number of workgroups per problem strides over the
// hypotheses the device-side count names.  blockIdx.x < gather_blocks: record copies of the problems on the fp32 / exact
// scorers; above: fp16 operand blocks of the problems on the matrix-core scorers.
__global__ __launch_bounds__(256) void k_gather_shadow16_g(const GroupArgs *ga, uint32_t gather_blocks) {
    const GroupArgs &g = ga[blockIdx.z];
    if (!g.active)
        return;
    const uint64_t cap = (uint64_t)g.comp.B * (uint64_t)g.comp.maxm;
    const uint32_t H = g.comp.ctl->num_hyp;
    if (blockIdx.x < gather_blocks) {
        if (g.comp.s16.out) // (the matrix-core scorer reads the records themselves)
            return;
        const uint64_t end = std::min<uint64_t>(cap, H) * 12u;
        for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < end; t += (uint64_t)gather_blocks * 256)
            gather_one(g.comp.ctl, g.comp.slots, g.comp.models, g.comp.shadow, g.comp.compact64, t);
        return;
    }
    if (!g.comp.s16.out)
        return;
    const uint32_t nb = gridDim.x - gather_blocks;
    const uint32_t k0 = (blockIdx.x - gather_blocks) * 256 + threadIdx.x;
    if (g.comp.s16.sampson == 2) {
        const uint32_t end = (uint32_t)std::min<uint64_t>((cap + 7u) & ~7ull, ((uint64_t)H + 7u) & ~7ull);
        for (uint32_t k = k0; k < end; k += nb * 256)
            hom16_one(k, H, g.comp.slots, g.comp.models, g.comp.s16.thr, static_cast<uint4 *>(g.comp.s16.out));
        return;
    }
    if (g.comp.s16.sampson) {
        const uint32_t capp = (uint32_t)std::min<uint64_t>(cap + kSampson16Pad, 0xffffff00ull);
        const uint32_t end = min(capp, H + (uint32_t)kSampson16Pad);
        for (uint32_t k = k0; k < end; k += nb * 256)
            sampson16_one(k, H, capp, g.comp.slots, g.comp.models, static_cast<uint4 *>(g.comp.s16.out));
        return;
    }
    // (shadow16_one fills the last group of 32 up and ignores everything behind it)
    const uint32_t end = (uint32_t)std::min<uint64_t>((cap + 31u) & ~31ull, ((uint64_t)H + 31u) & ~31ull);
    const uint32_t *slots = g.comp.slots;
    const double *models = g.comp.models;
    for (uint32_t k = k0; k < end; k += nb * 256)
        shadow16_one(k, H,
                     [&](uint32_t kk) { return reinterpret_cast<const float *>(models + (size_t)slots[kk] * kModelStride + kShadowOff); },
                     g.comp.s16.c16, g.comp.s16.thr, static_cast<uint4 *>(g.comp.s16.out));
}

// ------------------------------------------------------------------------------------ front-end pre-processing
__device__ __forceinline__ void prepare_body(const double *__restrict__ a_raw, const double *__restrict__ b_raw,
                                             uint32_t n, const PrepareArgs &g, double *__restrict__ soa,
                                             unsigned long long *absmax_bits) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    double m = 0.0;
    if (i < n) {
        double a0 = a_raw[2 * (size_t)i], a1 = a_raw[2 * (size_t)i + 1];
        if (g.mode == 2) {
            if (g.centred)
                a0 -= g.c1x, a1 -= g.c1y;
            a0 /= g.scale, a1 /= g.scale;
        } else {
            double u, v;
            camera_unproject(g.cam1, a0, a1, u, v);
            a0 = u, a1 = v;
        }
        soa[i] = a0;
        soa[(size_t)n + i] = a1;
        const double f0 = fabs(a0), f1 = fabs(a1);
        m = (f0 > m || f0 != f0) ? f0 : m; // NaN propagates (and disables the pre-filter), like the host loop
        m = (f1 > m || f1 != f1) ? f1 : m;
        if (g.mode == 0) {
            for (int d = 0; d < 3; ++d)
                soa[(size_t)(2 + d) * n + i] = b_raw[3 * (size_t)i + d];
        } else {
            double b0 = b_raw[2 * (size_t)i], b1 = b_raw[2 * (size_t)i + 1];
            if (g.mode == 2) {
                if (g.centred)
                    b0 -= g.c2x, b1 -= g.c2y;
                b0 /= g.scale, b1 /= g.scale;
            } else {
                double u, v;
                camera_unproject(g.cam2, b0, b1, u, v);
                b0 = u, b1 = v;
            }
            soa[(size_t)2 * n + i] = b0;
            soa[(size_t)3 * n + i] = b1;
        }
    }
    // non-negative doubles (and NaN, whose pattern is above +inf) order like their bit patterns
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bits, off, 64);
        bits = o > bits ? o : bits;
    }
    if ((threadIdx.x & 63) == 0 && bits && absmax_bits)
        atomicMax(absmax_bits, bits);
}
__global__ __launch_bounds__(256) void k_prepare(const double *__restrict__ a_raw, const double *__restrict__ b_raw,
                                                 uint32_t n, PrepareArgs g, double *__restrict__ soa,
                                                 unsigned long long *absmax_bits) {
    prepare_body(a_raw, b_raw, n, g, soa, absmax_bits);
}
__global__ __launch_bounds__(256) void k_prepare_g(const PrepareGroupArgs *pa) {
    const PrepareGroupArgs &g = pa[blockIdx.z];
    if (blockIdx.x * 256u >= g.n)
        return;
    prepare_body(g.a_raw, g.b_raw, g.n, g.args, g.soa, g.absmax_bits);
}

// Un-distortion as a stage of its own (BASELINE config 3: "OPENCV camera model" in front of the homography / 7-point
// estimators, which take no camera): pixel -> Camera::unproject (misc/camera_models.cc:1025-1032, the iterative
// OPENCV inverse :972-990) -> pixel of the distortion-free camera with the same focal lengths and principal point.
__global__ __launch_bounds__(256) void k_undistort(const double *__restrict__ in, uint32_t n, CameraParams cam, double fx,
                                                   double fy, double cx, double cy, double *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    double u, v;
    camera_unproject(cam, in[2 * (size_t)i], in[2 * (size_t)i + 1], u, v);
    out[2 * (size_t)i] = fx * u + cx;
    out[2 * (size_t)i + 1] = fy * v + cy;
}
// Diagnostic: the device's scalar math as the kernels call it, element-wise (pl_debug_device_math)
__global__ __launch_bounds__(256) void k_device_math(int fn, const double *__restrict__ x, uint32_t n, double *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const double v = x[i];
    double r;
    switch (fn) {
    case 0: r = lm_cube(v); break; // the cube of the Nielsen update (pl_refine.h)
    case 1: r = sqrt(v); break;
    case 2: r = 1.0 / v; break;
    case 3: r = pl_cbrt(v); break;
    case 4: r = pl_cos(v); break;
    case 5: r = pl_sin(v); break;
    case 6: r = pl_acos(v); break;
    default: r = v; break;
    }
    out[i] = r;
}
hipError_t launch_device_math(int fn, const double *x, uint32_t n, double *out, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    k_device_math<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(fn, x, n, out);
    return hipGetLastError();
}

hipError_t launch_undistort(const double *in, uint32_t n, const CameraParams &cam, double fx, double fy, double cx,
                            double cy, double *out, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    k_undistort<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(in, n, cam, fx, fy, cx, cy, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ launchers
hipError_t launch_shadow16(const uint32_t *num_hyp, const float *shadow_compact, uint32_t hyp_capacity, float g16,
                           float c16, float thr, void *shadow16, hipStream_t stream) {
    const uint32_t cap8 = (hyp_capacity + 31u) & ~31u;
    if (cap8 == 0)
        return hipSuccess;
    k_shadow16<<<dim3((cap8 + 255) / 256), dim3(256), 0, stream>>>(num_hyp, shadow_compact, cap8, g16, c16, thr,
                                                                   static_cast<uint4 *>(shadow16));
    return hipGetLastError();
}

hipError_t launch_prepare(const double *a_raw, const double *b_raw, uint32_t n, const PrepareArgs &args, double *soa,
                          unsigned long long *absmax_bits, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    k_prepare<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(a_raw, b_raw, n, args, soa, absmax_bits);
    return hipGetLastError();
}

hipError_t launch_sample_positions(int K, uint64_t seed, uint64_t pos_base, uint64_t N, uint32_t B, uint32_t M,
                                   uint8_t *delta, uint64_t *flagbits, uint32_t *positions, BatchCtl *ctl,
                                   uint32_t zero_words, hipStream_t stream) {
    uint32_t *const zero = reinterpret_cast<uint32_t *>(ctl);
    const dim3 grid((M + 255) / 256), block(256);
    switch (K) {
    case 3:
        k_sample_delta<3><<<grid, block, 0, stream>>>(seed, pos_base, N, M, delta, flagbits, zero, zero_words);
        break;
    case 4:
        k_sample_delta<4><<<grid, block, 0, stream>>>(seed, pos_base, N, M, delta, flagbits, zero, zero_words);
        break;
    case 5:
        k_sample_delta<5><<<grid, block, 0, stream>>>(seed, pos_base, N, M, delta, flagbits, zero, zero_words);
        break;
    case 7:
        k_sample_delta<7><<<grid, block, 0, stream>>>(seed, pos_base, N, M, delta, flagbits, zero, zero_words);
        break;
    default:
        return hipErrorInvalidValue;
    }
    if (orbit_small_build(M, N, K))
        k_sample_orbit<true><<<dim3(1), dim3(1024), 0, stream>>>(delta, flagbits, M, K, B, pos_base, positions, ctl);
    else
        k_sample_orbit<false><<<dim3(1), dim3(1024), 0, stream>>>(delta, flagbits, M, K, B, pos_base, positions, ctl);
    return hipGetLastError();
}

hipError_t launch_compact2(const uint32_t *num_models, uint32_t B, int maxm, uint32_t *blk_tot, bool counted, uint32_t *slots,
                           uint32_t *offsets, const double *models, float *shadow_compact, double *compact64,
                           BatchCtl *ctl, const Shadow16Params &s16, hipStream_t stream) {
    const uint32_t nb = (B + 1023) / 1024;
    if (!counted)
        k_count_blocks<<<dim3(nb), dim3(1024), 0, stream>>>(num_models, B, blk_tot);
    k_compact2<<<dim3(nb), dim3(1024), 0, stream>>>(num_models, B, maxm, blk_tot, slots, offsets, models, shadow_compact,
                                                    compact64, ctl);
    if (s16.out && s16.sampson == 2) {
        const uint32_t cap = (uint32_t)std::min<uint64_t>(((uint64_t)B * (uint64_t)maxm + 7u) & ~7ull, 0xffffff00ull);
        k_hom16<<<dim3(std::min<uint32_t>((cap + 255) / 256, 1024u)), dim3(256), 0, stream>>>(ctl, slots, models, cap, s16.thr,
                                                                                          static_cast<uint4 *>(s16.out));
    } else if (s16.out && s16.sampson) {
        const uint32_t cap = (uint32_t)std::min<uint64_t>((uint64_t)B * (uint64_t)maxm + kSampson16Pad, 0xffffff00ull);
        k_sampson16<<<dim3((cap + 255) / 256), dim3(256), 0, stream>>>(ctl, slots, models, cap, static_cast<uint4 *>(s16.out));
    } else if (s16.out) {
        // matrix-core scorer: its fp16 operand blocks are built straight from the records, and its exact pass reads the
        // fp64 models from the records as well - no hypothesis-ordered copies
        const uint32_t cap8 = (uint32_t)(((uint64_t)B * (uint64_t)maxm + 31u) & ~31ull);
        k_gather_shadow16<<<dim3((cap8 + 255) / 256), dim3(256), 0, stream>>>(ctl, slots, models, nullptr, nullptr, 0u, cap8,
                                                                            s16.g16, s16.c16, s16.thr,
                                                                            static_cast<uint4 *>(s16.out));
    } else if (shadow_compact && compact64) {
        const uint64_t threads = (uint64_t)B * (uint64_t)maxm * 12u; // capacity; lanes beyond num_hyp return at once
        k_gather_models<<<dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream>>>(ctl, slots, models,
                                                                                          shadow_compact, compact64);
    }
    return hipGetLastError();
}

hipError_t launch_sampson16(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t capacity, void *out,
                            hipStream_t stream) {
    const uint32_t cap = capacity + (uint32_t)kSampson16Pad;
    k_sampson16<<<dim3((cap + 255) / 256), dim3(256), 0, stream>>>(ctl, slots, models, cap, static_cast<uint4 *>(out));
    return hipGetLastError();
}

hipError_t launch_hom16(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t capacity, float thr, void *out,
                        hipStream_t stream) {
    const uint32_t cap = (capacity + 7u) & ~7u;
    k_hom16<<<dim3(std::min<uint32_t>((cap + 255) / 256, 1024u)), dim3(256), 0, stream>>>(ctl, slots, models, cap, thr,
                                                                                      static_cast<uint4 *>(out));
    return hipGetLastError();
}

hipError_t launch_gather_models(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t capacity,
                                float *shadow_compact, double *compact64, hipStream_t stream) {
    if (capacity == 0)
        return hipSuccess;
    const uint64_t threads = (uint64_t)capacity * 12u;
    k_gather_models<<<dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream>>>(ctl, slots, models, shadow_compact,
                                                                                      compact64);
    return hipGetLastError();
}
hipError_t launch_finalize(const FinalizeArgs &f, hipStream_t stream) {
    k_finalize2<<<dim3(kRecBlocks), dim3(256), 0, stream>>>(f, nullptr, nullptr);
    return hipGetLastError();
}

// ---- group launches (problem = blockIdx.z) ----
hipError_t launch_group_positions(int K, const GroupArgs *args, const GroupDims &d, hipStream_t stream) {
    const dim3 grid((d.max_M + 255) / 256, 1, d.G), block(256);
    switch (K) {
    case 3:
        k_sample_delta_g<3><<<grid, block, 0, stream>>>(args);
        break;
    case 4:
        k_sample_delta_g<4><<<grid, block, 0, stream>>>(args);
        break;
    case 5:
        k_sample_delta_g<5><<<grid, block, 0, stream>>>(args);
        break;
    case 7:
        k_sample_delta_g<7><<<grid, block, 0, stream>>>(args);
        break;
    default:
        return hipErrorInvalidValue;
    }
    if (orbit_small_build(d.max_M, d.min_n, K))
        k_sample_orbit_g<true><<<dim3(1, 1, d.G), dim3(1024), 0, stream>>>(args, K);
    else
        k_sample_orbit_g<false><<<dim3(1, 1, d.G), dim3(1024), 0, stream>>>(args, K);
    return hipGetLastError();
}
hipError_t launch_group_compact(const GroupArgs *args, const GroupDims &d, hipStream_t stream) {
    k_compact2_g<<<dim3((d.max_B + 1023) / 1024, 1, d.G), dim3(1024), 0, stream>>>(args);
    // bounded grids (see k_gather_shadow16_g): at most 1024 + 512 workgroups per problem, none where nobody needs them
    // (sized by ITERATIONS, not by record capacity: the lists hold 0.6 (5-point) ... 2.5 (7-point) hypotheses per iteration, the
    // kernels stride over the device-side count - by capacity a group of 228 5-point problems dispatched 180 k empty workgroups)
    const uint32_t gblocks = d.any_queue ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4u, ((uint64_t)d.max_B * 12u + 255) / 256), 1024u) : 0u;
    const uint32_t sblocks = d.any_mfma ? std::min<uint32_t>(std::max<uint32_t>(4u, (d.max_B * 2u + 255u) / 256u), 512u) : 0u;
    if (gblocks + sblocks)
        k_gather_shadow16_g<<<dim3(gblocks + sblocks, 1, d.G), dim3(256), 0, stream>>>(args, gblocks);
    return hipGetLastError();
}
hipError_t launch_group_finalize_records(const GroupArgs *args, const GroupDims &d, hipStream_t stream) {
    // chunks of >= 1024 hypotheses: a default-options problem has ~1.5 k of them - with the full 256 chunks per problem a group
    // of 228 problems dispatched 58 k almost empty workgroups per kernel (0.39 + 0.36 ms per step, r4b trace).  Both kernels cut
    // the list by gridDim.x; the running max / min scan is exact, so the candidates do not depend on the cut.
    const uint32_t nb = std::min<uint32_t>((uint32_t)kRecBlocks, std::max<uint32_t>(1u, (d.max_hcap + 1023u) / 1024u));
    k_finalize2_g<<<dim3(nb, 1, d.G), dim3(256), 0, stream>>>(args);
    k_records_g<<<dim3(nb, 1, d.G), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}
hipError_t launch_group_prepare(const PrepareGroupArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream) {
    if (G == 0 || max_n == 0)
        return hipSuccess;
    k_prepare_g<<<dim3((max_n + 255) / 256, 1, G), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}

hipError_t launch_finalize_records(const FinalizeArgs &f, const uint32_t *slots, const double *models,
                                   uint32_t *blk_max, double *blk_min, uint32_t init_max, double init_min,
                                   RecordMeta *rec_meta, double *rec_models, uint32_t rec_cap, BatchCtl *ctl,
                                   RecordMeta *host_meta, double *host_models, uint32_t host_cap, hipStream_t stream) {
    k_finalize2<<<dim3(kRecBlocks), dim3(256), 0, stream>>>(f, blk_max, blk_min);
    k_records<<<dim3(kRecBlocks), dim3(256), 0, stream>>>(f.num_hyp, f.count, f.score, slots, models, blk_max, blk_min,
                                                          init_max, init_min, rec_meta, rec_models, rec_cap, ctl,
                                                          host_meta, host_models, host_meta ? host_cap : 0u);
    return hipGetLastError();
}

} // namespace pl
