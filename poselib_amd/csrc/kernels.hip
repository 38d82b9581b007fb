// poselib_amd — HIP kernels for gfx950 (MI355X, CDNA4, wave64).  Build: hipcc --offload-arch=gfx950
// -O3 -std=c++17 -ffp-contract=off (fp64, no FMA contraction: residuals must round like the
// reference's SSE2 build so that inlier decisions are identical).
//
// Kernel inventory (one RANSAC batch = positions [pipeline.hip] -> generate -> compact/gather [pipeline.hip] ->
// score -> finalize/records [pipeline.hip] -> LM of the improving hypotheses -> re-score):
//   k_generate<EST>       one LANE per RANSAC iteration: counter-based sample draw (or explicit PROSAC sample), minimal
//                         solve entirely in registers (P3P / 5-pt / 7-pt / 4-pt H), model records (192 B incl. fp32
//                         shadow) written to HBM.  (reference: estimators/*::generate_models)
//   k_score_mfma<PG>      THE hot kernel (absolute pose): streaming scorer of the batched main loop - conservative
//                         pre-filter on the matrix cores: the four half-planes of the reprojection test as linear forms
//                         of fp16 hi/lo operands (v_mfma_f32_32x32x16_f16, 16 hypotheses x 32 correspondences per pair
//                         of instructions), survivors queued in LDS and evaluated exactly in fp64 by full wavefronts.
//                         (reference: utils.cc compute_msac_score)
//   k_score_mfma2<EST,PG> the same for the Sampson scores (relative pose, fundamental matrix) on coordinates bounded by
//                         8: the bilinear form b'Fa and the quadratic forms Cx + Cy out of the matrix pipe.
//   k_score_queue<EST,P>  the same with an fp32 pre-filter on the vector ALU: homography score, and the fallback of the
//                         others (small N, thresholds / coordinates outside the matrix-core forms' range).
//   k_score_seq<EST>      the MSAC score in the reference's summation order for every model a decision is taken on
//                         (candidates of the record scan, refined / initial models): one workgroup per model.
//   k_lm<EST>             Levenberg-Marquardt refinement, ONE workgroup (8 wavefronts) per task, the whole LM loop on
//                         device (up to 256 correspondences: sums in the reference's order); k_lm2<EST>: the same spread
//                         over several workgroups per task, one launch per LM iteration (opt-in).
//                         (reference: bundle.cc + optim/lm_impl.h + optim/*.h)
//   k_*_g                 group forms of the batch kernels: the same bodies, problem index = blockIdx.z, arguments from
//                         a device table (pl_estimate_batch / pl_ransac_batch, driver_group.inc).
//   k_mask<EST>           final inlier mask (reference: utils.cc get_inliers*).
//   k_solve_batch<EST>    the bare minimal solvers, one lane per problem.
// Exact arithmetic (fp64, reference association order) never runs on the matrix cores; only the filters' linear and
// quadratic forms do.
#include "pl_kernels.h"
#ifndef PL_XCD_MAP
#define PL_XCD_MAP 1
#endif
#include "pl_lm_chain.inc"
#include "pl_device.h"
#include <atomic>
#include "pl_prefilter.h"
#include <cstdlib>
#include "pl_sampler.h"
#include "pl_solver_h4.h"
#include "pl_solver_p3p.h"
#include "pl_solver_rel.h"

namespace pl {

// Wave-uniform data produced by an EARLIER kernel (model records, hypothesis lists) is read through the
// constant address space so that the compiler emits scalar loads (s_load_*, SGPR destination, scalar cache)
// instead of 64 identical vector loads — the kernels also store to global memory, which otherwise makes
// LLVM fall back to vector loads.
typedef const float __attribute__((address_space(4))) *uniform_f32_ptr;
typedef const double __attribute__((address_space(4))) *uniform_f64_ptr;
typedef const uint32_t __attribute__((address_space(4))) *uniform_u32_ptr;
__device__ __forceinline__ uniform_f32_ptr as_uniform(const float *p) { return (uniform_f32_ptr)(uintptr_t)p; }
__device__ __forceinline__ uniform_f64_ptr as_uniform(const double *p) { return (uniform_f64_ptr)(uintptr_t)p; }
__device__ __forceinline__ uniform_u32_ptr as_uniform(const uint32_t *p) { return (uniform_u32_ptr)(uintptr_t)p; }

// ------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}
// Wave reduction through the DPP cross-lane paths (no LDS round trips as in __shfl_xor): quads, half rows, rows of
// 16, then row broadcasts; the total is read from lane 63.  The pairing differs from wave_sum's xor butterfly, so
// the two agree to rounding only - the LM normal equations use this one, the MSAC scores keep wave_sum.
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ double dpp_move(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v); // row_half_mirror
    v += dpp_move<0x140>(v); // row_mirror: every lane holds the sum of its row of 16
    v += dpp_move<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3 (other rows add 0)
    v += dpp_move<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// The sums of the exact pass when the waiting pairs belong to several hypotheses: the pairs of one hypothesis are ONE run of
// lanes (the queues are filled hypothesis by hypothesis), lanes past the last pair carry a key no hypothesis has.  Run starts
// from one ballot (key against the key one lane below: DPP wave_shr:1), a lane's distance to its run's start replaces the
// key comparison of every scan step, the inlier count of a run is a popcount of the ballot - and the segmented inclusive scan
// of the values runs over the DPP paths (Hillis-Steele inside the rows of 16, then the row broadcasts), not over 26
// ds_bpermute round trips as in rounds 1 - 5: a drain is a dependency chain, not work, and this was the longest link of it
// (profiles/r06_score_phases.md).  The association of the additions differs from the shuffle form's: streaming scores are
// compared with a 1e-9 margin and every candidate is re-scored in the reference's order (k_finalize2 / k_score_seq).
__device__ __forceinline__ void add_run_totals(double v, uint64_t inmask, uint32_t g, bool act, int lane, double *acc_s, uint32_t *acc_c) {
    const uint32_t gprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x138, 0xf, 0xf, false); // wave_shr:1
    const uint64_t heads = __builtin_amdgcn_ballot_w64(lane == 0 || gprev != g);
    const uint64_t upto = ~0ull >> (63 - lane); // lanes 0 .. mine
    const int start = 63 - __clzll((long long)(heads & upto));
    const uint32_t dist = (uint32_t)(lane - start);
    double t;
    t = dpp_move<0x111>(v), v += dist >= 1u ? t : 0.0; // row_shr:1 (lanes whose source is outside their row of 16 receive 0)
    t = dpp_move<0x112>(v), v += dist >= 2u ? t : 0.0;
    t = dpp_move<0x114>(v), v += dist >= 4u ? t : 0.0;
    t = dpp_move<0x118>(v), v += dist >= 8u ? t : 0.0;
    // the part of the run in the rows below: lane 15 resp. 47 into rows 1 resp. 3, then lane 31 into rows 2 and 3
    t = dpp_move<0x142, 0xa>(v), v += dist > (uint32_t)(lane & 15) ? t : 0.0;
    t = dpp_move<0x143, 0xc>(v), v += (lane >= 32 && dist >= (uint32_t)(lane - 31)) ? t : 0.0;
    const bool tail = act && (lane == 63 || ((heads >> (lane + 1)) & 1ull));
    const uint32_t c = (uint32_t)__popcll(inmask & upto & (~0ull << start));
    if (tail && c) {
        acc_s[g] += v;
        acc_c[g] += c;
    }
}
// Inclusive prefix sum over the 64 lanes through the DPP paths (Hillis-Steele inside each row of 16 with zero fill,
// then the row totals are broadcast downwards): no LDS round trips.
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ uint32_t dpp_shift_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    v += dpp_shift_u32<0x111>(v); // row_shr:1
    v += dpp_shift_u32<0x112>(v); // row_shr:2
    v += dpp_shift_u32<0x114>(v); // row_shr:4
    v += dpp_shift_u32<0x118>(v); // row_shr:8
    v += dpp_shift_u32<0x142, 0xa>(v); // row_bcast:15 -> rows 1, 3
    v += dpp_shift_u32<0x143, 0xc>(v); // row_bcast:31 -> rows 2, 3
    return v;
}
// ------------------------------------------------------------------------------------ generate
// P3P in two halves (pl_solver_p3p.h).  A wavefront = 64 iterations.  Every lane runs the first half on its own sample and
// appends its <= 4 candidate depth triples to the wave's list in LDS; then the lanes take the CANDIDATES of the list, 64 at
// a time - polish, R, t, record - fetching the sample's data from the lane that owns it through the cross-lane network
// (ds_bpermute: no LDS memory).  1.3 of the 4 candidate slots are filled on average, and a wavefront executes a slot as soon as
// one lane fills it: run per lane the second half costs four rounds, run on the list it costs two (83 candidates on average).
// Records go to (iteration, solution index) as before: the outcome is bit-identical, only the lane that computes a solution
// changes.  (Workgroup = one wavefront: the barrier below only orders the LDS traffic.)
__device__ __forceinline__ double lane_fetch(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ Vec3 lane_fetch(Vec3 v, int src) {
    return v3(lane_fetch(v.x, src), lane_fetch(v.y, src), lane_fetch(v.z, src));
}
__device__ __forceinline__ int generate_abs_dense(const GenerateArgs &g, uint32_t it, bool live, const Vec3 *xb, const Vec3 *Xp,
                                                  uint32_t &n_nan) {
    __shared__ uint8_t s_src[256];      // candidate -> owning lane | solution index << 6
    __shared__ double s_cand[3][256];   // its depths
    const int lane = threadIdx.x & 63;
    P3PFront f;
    double cand[4][3];
    int n = 0;
    if (live)
        n = p3p_front(xb[0], xb[1], xb[2], Xp[0], Xp[1], Xp[2], f, cand);
    const uint32_t incl = wave_scan_u32((uint32_t)n);
    const uint32_t first = incl - (uint32_t)n;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m < n) {
            s_src[first + m] = (uint8_t)(lane | (m << 6));
            s_cand[0][first + m] = cand[m][0];
            s_cand[1][first + m] = cand[m][1];
            s_cand[2][first + m] = cand[m][2];
        }
    __syncthreads();
    const uint32_t it0 = it - (uint32_t)lane; // the wave's first iteration
    for (uint32_t base = 0; base < total; base += 64u) { // (wave-uniform trip count)
        const uint32_t e = base + (uint32_t)lane;
        const bool act = e < total;
        const uint32_t src = act ? (uint32_t)s_src[e] : 0u;
        const int owner = (int)(src & 63u), m = (int)(src >> 6);
        P3PFront o; // the owner's sample (every lane takes part in the exchange)
        o.x0 = lane_fetch(f.x0, owner), o.x1 = lane_fetch(f.x1, owner), o.x2 = lane_fetch(f.x2, owner);
        o.X0 = lane_fetch(f.X0, owner);
#pragma unroll
        for (int k = 0; k < 9; ++k)
            o.XX.m[k] = lane_fetch(f.XX.m[k], owner);
        o.a01 = lane_fetch(f.a01, owner), o.a02 = lane_fetch(f.a02, owner), o.a12 = lane_fetch(f.a12, owner);
        o.m01 = lane_fetch(f.m01, owner), o.m02 = lane_fetch(f.m02, owner), o.m12 = lane_fetch(f.m12, owner);
        if (act) {
            Mat3 R;
            Vec3 t;
            p3p_back(o, s_cand[0][e], s_cand[1][e], s_cand[2][e], R, t);
            double *rec = g.models + ((size_t)(it0 + (uint32_t)owner) * g.slots_per_iter + (uint32_t)m) * kModelStride;
            n_nan += store_pose_model(rec, R, t, false) ? 1u : 0u;
        }
    }
    __syncthreads(); // (the list is rewritten by the next call - solver batches run several per workgroup)
    return n;
}

template <int EST> __device__ __forceinline__ uint32_t generate_one(const GenerateArgs &g, uint32_t it, uint32_t &n_nan) {
    constexpr int K = sample_size(EST);
    constexpr int MAXM = max_models(EST);
    uint32_t idx[K];
    if (g.samples) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            idx[k] = g.samples[(size_t)it * K + k];
    } else {
        draw_sample<K>(g.seed, g.pos_base + g.positions[it], g.pts.n, idx);
    }
    (void)MAXM;
    double *rec = g.models + (size_t)it * g.slots_per_iter * kModelStride;
    int n = 0;
    if constexpr (EST == EST_ABS) {
        Vec3 xb[3], Xp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xb[k] = bearing(g.pts.a[0][idx[k]], g.pts.a[1][idx[k]]);
            Xp[k] = v3(g.pts.a[2][idx[k]], g.pts.a[3][idx[k]], g.pts.a[4][idx[k]]);
        }
        n = p3p_emit(xb[0], xb[1], xb[2], Xp[0], Xp[1], Xp[2], [&](int m, const Mat3 &R, const Vec3 &t) {
            n_nan += store_pose_model(rec + m * kModelStride, R, t, false) ? 1u : 0u;
        }); // (the generator kernels take generate_abs_wave instead: the second half on full wavefronts of candidates)
    } else {
        Vec3 b1[K], b2[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            b1[k] = bearing(g.pts.a[0][idx[k]], g.pts.a[1][idx[k]]);
            b2[k] = bearing(g.pts.a[2][idx[k]], g.pts.a[3][idx[k]]);
        }
        if constexpr (EST == EST_HOM) {
            Mat3 H;
            n = homography_4pt(b1, b2, H, true);
            if (n)
                n_nan += store_matrix_model(rec, H) ? 1u : 0u;
        } else if constexpr (EST == EST_FUND) {
            n = relpose_7pt_records(b1, b2, rec, g.real_focal_check != 0);
            for (int m = 0; m < n; ++m)
                n_nan += record_is_nan(rec + m * kModelStride);
        } else {
            n = relpose_5pt_records(b1, b2, rec, (int)g.slots_per_iter);
            if (n > (int)g.slots_per_iter) {
                g.ctl->gen_overflow = 1;
                n = 0;
            }
            for (int m = 0; m < n; ++m)
                n_nan += record_is_nan(rec + m * kModelStride);
        }
    }
    g.num_models[it] = (uint32_t)n;
    return (uint32_t)n;
}
// absolute pose: all 64 lanes of the wavefront call (collective second half); lanes past the last iteration carry no sample
__device__ __forceinline__ uint32_t generate_abs_wave(const GenerateArgs &g, uint32_t it, uint32_t &n_nan) {
    const bool live = it < g.num_iters;
    Vec3 xb[3], Xp[3];
    if (live) {
        uint32_t idx[3];
        sample_of_iteration<3>(g, it, idx);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xb[k] = bearing(g.pts.a[0][idx[k]], g.pts.a[1][idx[k]]);
            Xp[k] = v3(g.pts.a[2][idx[k]], g.pts.a[3][idx[k]], g.pts.a[4][idx[k]]);
        }
    }
    const int n = generate_abs_dense(g, it, live, xb, Xp, n_nan);
    if (live)
        g.num_models[it] = (uint32_t)n;
    return live ? (uint32_t)n : 0u;
}
template <int EST> __global__ __launch_bounds__(64) void k_generate(GenerateArgs g) {
    const uint32_t it = blockIdx.x * 64 + threadIdx.x;
    uint32_t n_nan = 0;
    uint32_t n;
    if constexpr (EST == EST_ABS)
        n = generate_abs_wave(g, it, n_nan);
    else
        n = (it < g.num_iters) ? generate_one<EST>(g, it, n_nan) : 0u;
    count_models_of_wave(g, it, n, n_nan); // one call site: the lanes past the last iteration take part with n = 0
}

template <int EST> __global__ __launch_bounds__(64) void k_generate_g(const GroupArgs *ga) {
    const GroupArgs &gg = ga[blockIdx.z];
    if (!gg.active || blockIdx.x * 64u >= gg.gen.num_iters)
        return;
    const GenerateArgs &g = gg.gen;
    const uint32_t it = blockIdx.x * 64 + threadIdx.x;
    uint32_t n_nan = 0;
    uint32_t n;
    if constexpr (EST == EST_ABS)
        n = generate_abs_wave(g, it, n_nan);
    else
        n = (it < g.num_iters) ? generate_one<EST>(g, it, n_nan) : 0u;
    count_models_of_wave(g, it, n, n_nan);
}

// Bare solver batch: one lane per minimal problem, AoS input exactly as the reference API takes it.
template <int EST> __global__ __launch_bounds__(64) void k_solve_batch(const double *in, uint32_t np, double *models,
                                                                       uint32_t *num_models) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= np)
        return;
    constexpr int K = sample_size(EST);
    constexpr int MAXM = max_models(EST);
    const double *p = in + (size_t)i * 6 * K;
    Vec3 a[K], b[K];
    for (int k = 0; k < K; ++k) {
        a[k] = v3(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
        b[k] = v3(p[3 * (K + k)], p[3 * (K + k) + 1], p[3 * (K + k) + 2]);
    }
    double *rec = models + (size_t)i * MAXM * kModelStride;
    int n = 0;
    if constexpr (EST == EST_ABS) {
        P3PSolution sol[4];
        n = p3p(a[0], a[1], a[2], b[0], b[1], b[2], sol);
        for (int m = 0; m < n; ++m)
            store_pose_model(rec + m * kModelStride, sol[m].R, sol[m].t, false);
    } else if constexpr (EST == EST_HOM) {
        Mat3 H;
        n = homography_4pt(a, b, H, true);
        if (n)
            store_matrix_model(rec, H);
    } else if constexpr (EST == EST_FUND) {
        n = relpose_7pt_records(a, b, rec, false);
    } else {
        n = relpose_5pt_records(a, b, rec);
    }
    num_models[i] = (uint32_t)n;
}

// 5-point essential matrices without the motion decomposition (poselib.essential_matrix_5pt).
__global__ __launch_bounds__(64) void k_solve_essential(const double *in, uint32_t np, double *models,
                                                        uint32_t *num_models) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= np)
        return;
    const double *p = in + (size_t)i * 30;
    Vec3 a[5], b[5];
    for (int k = 0; k < 5; ++k) {
        a[k] = v3(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
        b[k] = v3(p[3 * (5 + k)], p[3 * (5 + k) + 1], p[3 * (5 + k) + 2]);
    }
    Mat3 E[10];
    const int n = essential_5pt(a, b, E);
    for (int m = 0; m < n; ++m)
        store_matrix_model(models + ((size_t)i * 10 + m) * kModelStride, E[m]);
    num_models[i] = (uint32_t)n;
}

// ------------------------------------------------------------------------------------ score
template <int EST>
__device__ __forceinline__ bool eval_point(const double *M, const double *pt, double thr2, double &r2) {
    if constexpr (EST == EST_ABS)
        return reproj_inlier(M, pt[0], pt[1], pt[2], pt[3], pt[4], thr2, r2);
    else if constexpr (EST == EST_REL)
        return sampson_pose_inlier(M, pt[0], pt[1], pt[2], pt[3], thr2, r2);
    else if constexpr (EST == EST_FUND)
        return sampson_inlier(M, pt[0], pt[1], pt[2], pt[3], thr2, r2);
    else
        return homography_inlier(M, pt[0], pt[1], pt[2], pt[3], thr2, r2);
}

#ifdef PL_SCALAR_ABS_FILTER
constexpr bool kPackedAbsFilter = false;
#else
constexpr bool kPackedAbsFilter = true; // v_pk_fma_f32 pairs in the absolute-pose filter
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f bc(float s) { return v2f{s, s}; }

// ---- work distribution of the streaming scorers ---------------------------------------------------------------------
// The waves that hold the same chunk of correspondences (all workgroups of the launch with that chunk index) share the
// hypothesis stream by a fixed stride: workgroup b of B owns the units 8 b + j + 8 B r (j < 8 waves, r = 0, 1, ...) and
// its waves take them one at a time from a counter in LDS, so they finish together.  Units are 64 hypotheses while
// whole rounds remain (every wave gets the same number of them), then the remainder is cut into units of 32 or 16 so
// that the last round is short as well - with 2025 groups for 384 waves (config 1) contiguous ranges per workgroup
// cost 6 rounds, this costs 5.5; with 918 for 432 (config 2) 2.25 instead of 3.  (A counter in global memory shared by
// the waves of all XCDs was measured: 2.2x SLOWER - contended device-scope atomics.)  Which wave evaluates a unit has
// no influence on its results.
template <uint32_t MINSUB = 16u>
__device__ __forceinline__ bool unit_of_ticket(uint32_t t, uint32_t H, uint32_t waves_per_chunk, uint32_t &kb, uint32_t &gn) {
    const uint32_t G = (H + 63u) / 64u;
    const uint32_t full = (G / waves_per_chunk) * waves_per_chunk; // 64-hypothesis units of the whole rounds
    if (t < full) {
        kb = t * 64u;
        gn = min(64u, H - kb);
        return true;
    }
    const uint32_t rest = G - full; // < waves_per_chunk groups left
    const uint32_t sub = (MINSUB <= 16u && rest * 4u <= waves_per_chunk) ? 16u : (rest * 2u <= waves_per_chunk) ? 32u : 64u;
    kb = full * 64u + (t - full) * sub;
    if (kb >= H)
        return false;
    gn = min(sub, H - kb);
    return true;
}

// ---- deferred exact evaluation ----------------------------------------------------------------------------------
// k_score_queue: the scorer of the batched main loop.  One wavefront = 64*P register-resident correspondences (fp32
// copies + bound terms) x a stream of hypotheses; the four waves of a workgroup share the same correspondences
// (fp64 copies in LDS) and split the hypotheses between them, so nothing in the loop needs a workgroup barrier.
//   pass A   conservative fp32 pre-filter (pl_prefilter.h), results = wave masks in SGPRs;
//   queue    the (hypothesis, correspondence) pairs that survive are appended to a wave-private LDS ring
//            (ballot + mbcnt compaction) instead of being evaluated on the spot - a wrong hypothesis leaves a
//            handful of survivors, and evaluating a 64-lane slot for one of them would cost as much as pass A;
//   drain    whenever 64 pairs are waiting (and at the end of each group of 64 hypotheses) every lane takes one
//            pair: its fp64 correspondence from LDS, its fp64 model from the compact stream, the exact expression
//            of pl_score.h.  Pairs are ordered by (hypothesis, correspondence), so a segmented inclusive scan over
//            the lanes (keys = hypothesis) yields per-hypothesis sums in a fixed order; the last lane of every
//            segment adds them to the wave's per-hypothesis accumulators in LDS.
// Everything is deterministic (no atomics); the summation tree differs from the non-streaming kernels, i.e. scores
// agree with them to rounding, counts exactly.
constexpr int kQueueCap = 512; // >= 63 + 64 * 6 entries can be waiting at most
constexpr int kQueueThreads = 512; // 8 wavefronts share one chunk of correspondences

template <int EST, int P>
__device__ __forceinline__ void score_queue_body(const PointSet &pts, const float *__restrict__ shadow,
                                                 const double *__restrict__ compact64,
                                                 const uint32_t *__restrict__ num_hyp_ptr, uint32_t hyp_capacity,
                                                 double thr2, const PrefilterArgs &pf, uint32_t *__restrict__ part_count,
                                                 double *__restrict__ part_score, uint32_t slice, uint32_t chunk,
                                                 uint32_t nslices) {
    constexpr int kWaves = kQueueThreads / 64;
    constexpr int ND = point_doubles(EST);
    constexpr int NB = 1; // bound terms per point
    constexpr int NPW = 64 * P;                                         // correspondences per chunk
    __shared__ double s_pts[ND][NPW];
    // relative pose: the unit bearings of the cheirality test (utils.cc:183-185) depend on the correspondence only -
    // computed once per chunk instead of once per exact evaluation (2 square roots + 6 divisions, 40 % of a drain)
    constexpr int NBR = (EST == EST_REL) ? 6 : 1;
    __shared__ double s_bear[NBR][(EST == EST_REL) ? NPW : 1];
    __shared__ uint16_t s_queue[kWaves][kQueueCap]; // entries: hypothesis of the group << 9 | correspondence of the chunk
    __shared__ double s_acc_s[kWaves][64];
    __shared__ uint32_t s_acc_c[kWaves][64];
    __shared__ uint32_t s_next_unit;
    const int lane = threadIdx.x & 63;
    // readfirstlane: the wave index is uniform, and the compiler has to know it for the scalar (s_load) shadow stream
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    float pf32[P][ND]; // the correspondences in fp32
    float bnd[P][NB];  // per-point bound terms of the pre-filter
    uint64_t vmask[P]; // lanes of slot p that hold a real correspondence (wave-uniform)
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t i = chunk * NPW + p * 64 + lane;
        const bool valid = i < pts.n;
        const uint32_t ic = valid ? i : 0u;
        vmask[p] = __builtin_amdgcn_ballot_w64(valid);
        double x[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            x[d] = pts.a[d][ic];
            pf32[p][d] = (float)x[d];
            if (wave == 0)
                s_pts[d][p * 64 + lane] = x[d];
        }
        if constexpr (EST == EST_REL) {
            if (wave == 1 + (p % (kWaves - 1))) {
                const Vec3 u1 = bearing(x[0], x[1]), u2 = bearing(x[2], x[3]);
                s_bear[0][p * 64 + lane] = u1.x, s_bear[1][p * 64 + lane] = u1.y, s_bear[2][p * 64 + lane] = u1.z;
                s_bear[3][p * 64 + lane] = u2.x, s_bear[4][p * 64 + lane] = u2.y, s_bear[5][p * 64 + lane] = u2.z;
            }
        }
        if constexpr (EST == EST_ABS) {
            bnd[p][0] = pf_point_abs(x[2], x[3], x[4], pf.gx);
        } else {
            float nanb, nsq, nanb_thr;
            pf_point_two_view(x[0], x[1], x[2], x[3], pf.thr, nanb, nsq, nanb_thr);
            if constexpr (EST == EST_HOM)
                bnd[p][0] = nanb_thr;
            else
                bnd[p][0] = pf_point_sampson_w(nanb, nsq, pf);
        }
    }
    if (threadIdx.x == 0)
        s_next_unit = 0;
    __syncthreads(); // the only workgroup barrier: fp64 correspondences are in LDS

    const uint32_t H = *as_uniform(num_hyp_ptr);
    const uniform_f32_ptr sh = as_uniform(shadow);
    uint16_t *const queue = s_queue[wave];
    double *const acc_s = s_acc_s[wave];
    uint32_t *const acc_c = s_acc_c[wave];

    const uint32_t waves_per_chunk = nslices * kWaves;
    auto request_ticket = [&]() -> uint32_t { // per-lane value; lane 0 holds the workgroup's next unit
        uint32_t t = 0;
        if (lane == 0) {
            const uint32_t k = atomicAdd(&s_next_unit, 1u);
            t = slice * kWaves + (k % kWaves) + (k / kWaves) * waves_per_chunk;
        }
        return t;
    };
    uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)request_ticket());
    uint32_t kb, gn;
    while (unit_of_ticket(ticket, H, waves_per_chunk, kb, gn)) {
        const uint32_t pending = request_ticket(); // the next unit's index travels while this one is evaluated
        acc_s[lane] = 0.0;
        acc_c[lane] = 0;
        uint32_t qhead = 0, qtail = 0; // wave-uniform ring positions

        auto drain = [&](uint32_t n) { // n <= 64 waiting pairs, one per lane
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool act = (uint32_t)lane < n;
            const uint32_t e = act ? (uint32_t)queue[(qhead + lane) & (kQueueCap - 1)] : 0xffffu;
            const uint32_t g = e >> 9, pi = act ? (e & 0x1ffu) : 0u;
            double x[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d)
                x[d] = s_pts[d][pi];
            const double *Mp = compact64 + (size_t)(kb + (act ? g : 0u)) * kModelDoubles;
            double M[kModelDoubles];
#pragma unroll
            for (int i = 0; i < kModelDoubles; ++i)
                M[i] = Mp[i];
            double r2;
            bool in;
            if constexpr (EST == EST_REL) {
                r2 = sampson_sq(M + kMatOff, x[0], x[1], x[2], x[3]);
                in = r2 < thr2;
                if (in) {
                    Quat q;
                    q.w = M[0], q.x = M[1], q.y = M[2], q.z = M[3];
                    in = check_cheirality(q, v3(M[4], M[5], M[6]), v3(s_bear[0][pi], s_bear[1][pi], s_bear[2][pi]),
                                          v3(s_bear[3][pi], s_bear[4][pi], s_bear[5][pi]), 0.01);
                }
                in = in && act;
            } else {
                in = eval_point<EST>(M, x, thr2, r2) && act;
            }
            double v = in ? r2 : 0.0;
            const uint64_t inmask = __builtin_amdgcn_ballot_w64(in);
            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
            if (inmask && !__builtin_amdgcn_ballot_w64(act && g != g0)) {
                // all waiting pairs belong to one hypothesis (the usual case when a good model's inliers arrive): a plain
                // wave sum over the DPP paths
                const double tot = wave_sum_dpp(v);
                if (lane == 0) {
                    acc_s[g0] += tot;
                    acc_c[g0] += (uint32_t)__popcll(inmask);
                }
            } else if (inmask) {
                // segmented inclusive scan; keys (hypothesis) ascend with the lane, so "same key `off` lanes below"
                // implies the whole stretch in between belongs to the segment
                add_run_totals(v, inmask, g, act, lane, acc_s, acc_c);
            }
            qhead += n;
        };

        auto step = [&](const float(&r)[15], uint32_t g) {
            if (__float_as_uint(r[13]) != 0u)
                return; // NaN model: zero inliers (pl_math.h store_shadow)
            uint64_t m[P];
            uint64_t any = 0;
            if (!pf.enabled) {
#pragma unroll
                for (int p = 0; p < P; ++p)
                    m[p] = vmask[p], any |= m[p];
            } else if constexpr (EST == EST_ABS && kPackedAbsFilter) {
                const float gt = pf_up(pf.gx * r[12]);
                // two points per packed fp32 instruction; the comparisons feed the ballots directly
#pragma unroll
                for (int p = 0; p + 1 < P; p += 2) {
                    const v2f X = {pf32[p][2], pf32[p + 1][2]}, Y = {pf32[p][3], pf32[p + 1][3]};
                    const v2f Z = {pf32[p][4], pf32[p + 1][4]};
                    const v2f x = {pf32[p][0], pf32[p + 1][0]}, y = {pf32[p][1], pf32[p + 1][1]};
                    const v2f w = {bnd[p][0], bnd[p + 1][0]};
                    const v2f z0 = pk_fma(bc(r[0]), X, pk_fma(bc(r[1]), Y, pk_fma(bc(r[2]), Z, bc(r[9]))));
                    const v2f z1 = pk_fma(bc(r[3]), X, pk_fma(bc(r[4]), Y, pk_fma(bc(r[5]), Z, bc(r[10]))));
                    const v2f z2 = pk_fma(bc(r[6]), X, pk_fma(bc(r[7]), Y, pk_fma(bc(r[8]), Z, bc(r[11]))));
                    const v2f a0 = pk_fma(-x, z2, z0);
                    const v2f a1 = pk_fma(-y, z2, z1);
                    const v2f W = w + bc(gt);
                    const v2f B = pk_fma(bc(pf.thr), z2, W);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const uint64_t far = __builtin_amdgcn_ballot_w64(fmaxf(fabsf(a0[e]), fabsf(a1[e])) > B[e]);
                        const uint64_t behind = __builtin_amdgcn_ballot_w64(z2[e] < -W[e]);
                        m[p + e] = vmask[p + e] & ~(far | behind);
                        any |= m[p + e];
                    }
                }
                if constexpr (P & 1) {
                    constexpr int p = P - 1;
                    const bool out = pf_abs_outlier(r, gt, pf.thr, pf32[p][0], pf32[p][1], pf32[p][2], pf32[p][3],
                                                    pf32[p][4], bnd[p][0]);
                    m[p] = vmask[p] & ~__builtin_amdgcn_ballot_w64(out);
                    any |= m[p];
                }
            } else if constexpr (EST == EST_ABS) {
                const float gt = pf_up(pf.gx * r[12]);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const bool out = pf_abs_outlier(r, gt, pf.thr, pf32[p][0], pf32[p][1], pf32[p][2], pf32[p][3],
                                                    pf32[p][4], bnd[p][0]);
                    m[p] = vmask[p] & ~__builtin_amdgcn_ballot_w64(out);
                    any |= m[p];
                }
            } else if constexpr (EST == EST_HOM) {
                const float gh = (32.f * kPfU) * r[14];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const bool out = pf_hom_outlier(r, gh, pf.thr, pf32[p][0], pf32[p][1], pf32[p][2], pf32[p][3],
                                                    bnd[p][0]);
                    m[p] = vmask[p] & ~__builtin_amdgcn_ballot_w64(out);
                    any |= m[p];
                }
            } else {
                const float gf = (16.f * kPfU) * r[14];
                const float gf2 = gf * gf;
                // two points per packed fp32 instruction, operation for operation pf_sampson_outlier: ONE comparison per
                // point (pl_prefilter.h), its result is the wave mask
#pragma unroll
                for (int p = 0; p + 1 < P; p += 2) {
                    const v2f a0 = {pf32[p][0], pf32[p + 1][0]}, a1 = {pf32[p][1], pf32[p + 1][1]};
                    const v2f b0 = {pf32[p][2], pf32[p + 1][2]}, b1 = {pf32[p][3], pf32[p + 1][3]};
                    const v2f w = {bnd[p][0], bnd[p + 1][0]};
                    const v2f Ea0 = pk_fma(bc(r[0]), a0, pk_fma(bc(r[1]), a1, bc(r[2])));
                    const v2f Ea1 = pk_fma(bc(r[3]), a0, pk_fma(bc(r[4]), a1, bc(r[5])));
                    const v2f Ea2 = pk_fma(bc(r[6]), a0, pk_fma(bc(r[7]), a1, bc(r[8])));
                    const v2f Eb0 = pk_fma(bc(r[0]), b0, pk_fma(bc(r[3]), b1, bc(r[6])));
                    const v2f Eb1 = pk_fma(bc(r[1]), b0, pk_fma(bc(r[4]), b1, bc(r[7])));
                    const v2f C = pk_fma(b0, Ea0, pk_fma(b1, Ea1, Ea2));
                    const v2f S = pk_fma(Eb1, Eb1, pk_fma(Eb0, Eb0, pk_fma(Ea1, Ea1, Ea0 * Ea0)));
                    const v2f L = C * C;
                    const v2f R = pk_fma(bc(pf.t1), S, bc(gf2) * w);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        m[p + e] = vmask[p + e] & ~__builtin_amdgcn_ballot_w64(L[e] > R[e]);
                        any |= m[p + e];
                    }
                }
                if constexpr (P & 1) {
                    constexpr int p = P - 1;
                    const bool out = pf_sampson_outlier(r, gf, pf.t1, pf32[p][0], pf32[p][1], pf32[p][2], pf32[p][3],
                                                        bnd[p][0]);
                    m[p] = vmask[p] & ~__builtin_amdgcn_ballot_w64(out);
                    any |= m[p];
                }
            }
            if (any) { // wave-uniform: append the survivors, ordered by (slot, lane)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (m[p]) {
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[p] >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)m[p], 0u));
                        if ((m[p] >> lane) & 1u)
                            queue[(qtail + below) & (kQueueCap - 1)] = (uint16_t)((g << 9) | (uint32_t)(p * 64 + lane));
                        qtail += (uint32_t)__popcll(m[p]);
                    }
                }
                while (qtail - qhead >= 64u)
                    drain(64u);
            }
        };
        auto fetch = [&](float(&r)[15], uint32_t g) {
            const uniform_f32_ptr sp = sh + (size_t)(kb + g) * 16;
#pragma unroll
            for (int i = 0; i < 15; ++i)
                r[i] = sp[i];
        };

        float ra[15], rb[15];
        fetch(ra, 0);
        for (uint32_t g = 0; g < gn; g += 2) {
            const bool two = g + 1 < gn;
            if (two)
                fetch(rb, g + 1);
            step(ra, g);
            if (two) {
                if (g + 2 < gn)
                    fetch(ra, g + 2);
                step(rb, g + 1);
            }
        }
        while (qtail != qhead)
            drain(min(64u, qtail - qhead));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((uint32_t)lane < gn) {
            const size_t o = (size_t)chunk * hyp_capacity + kb + lane;
            part_score[o] = acc_s[lane];
            part_count[o] = acc_c[lane];
        }
        ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)pending);
    }
}

template <int EST, int P>
__global__ __launch_bounds__(kQueueThreads) void k_score_queue(PointSet pts, const float *__restrict__ shadow,
                                                                const double *__restrict__ compact64,
                                                                const uint32_t *__restrict__ num_hyp_ptr,
                                                                uint32_t hyp_capacity, double thr2, PrefilterArgs pf,
                                                                uint32_t *__restrict__ part_count,
                                                                double *__restrict__ part_score,
                                                                uint32_t *__restrict__ tickets) {
    (void)tickets;
    score_queue_body<EST, P>(pts, shadow, compact64, num_hyp_ptr, hyp_capacity, thr2, pf, part_count, part_score,
                             blockIdx.x, blockIdx.y, gridDim.x);
}
template <int EST, int P> __global__ __launch_bounds__(kQueueThreads) void k_score_queue_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    if (!g.active || g.use_mfma || blockIdx.y >= g.chunks || blockIdx.x >= g.slices)
        return;
    const ScoreArgs &a = g.score;
    score_queue_body<EST, P>(a.pts, a.shadow, a.compact64, a.num_hyp, a.hyp_capacity, a.thr2, a.pf, a.part_count,
                             a.part_score, blockIdx.x, blockIdx.y, g.slices);
}

// ---- absolute pose: the pre-filter on the matrix cores ----------------------------------------------------------
// k_score_mfma: same contract and same exact pass as k_score_queue, but pass A comes out of the matrix pipe whole.  An
// inlier's residual vector is shorter than thr z_2, so its component along any direction is: three directions 120 degrees
// apart bound the inlier disc by a triangle (round 6; rounds 2 - 5: the axis-parallel square, four half-planes), and each
// half-plane is linear in sixteen numbers of the correspondence (X high / low, 1, slack | p X high / low, p, |p|; p = c x + s y),
// so three v_mfma_f32_32x32x16_f16 - one per direction, 32 hypotheses x 32 correspondences each - deliver the three signed
// distances of a pair, slack included (operand rows: k_shadow16 in pipeline.hip; bound derivation in pl_prefilter.h).  The
// vector ALU ORs the three sign bits of a pair (one v_or3) and shifts the result into a per-lane, per-hypothesis bit field over
// the point groups (v_alignbit): 2 instructions per pair (round 2: 3, round 1: 4); the triangle lets 1.3 x the square's pairs
// through to the exact pass.  After the PG tiles of a group of 32 hypotheses the bit fields are expanded into the wave's LDS
// queue, hypothesis by hypothesis (prefix sum over the lanes), so the drain's segmented scan sees every hypothesis as one
// run, exactly as in k_score_queue.
// Register layout of a tile (32 rows x 32 columns, 16 accumulator registers per lane): lane l = column l % 32, register v =
// hypothesis row 8 (v / 4) + 4 (l / 32) + v % 4 - the same for the three directions.
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
#ifndef PL_MFMA_THREADS
#define PL_MFMA_THREADS 512
#endif
#ifndef PL_ABS_TILE_UNROLL
#define PL_ABS_TILE_UNROLL 16 // k_score_mfma's loop over the point groups of a tile: unrolled completely (PG - 1 <= 9 iterations) - the LDS
                             // addresses of the operands become immediate offsets (3 of 35 vector instructions per group): 131.0 -> 126.5 us
                             // per launch (unrolled 3 times: 125.9 / 129.1; profiles/r06_score_phases.md)
#endif
#ifndef PL_ABS_SCHED
#define PL_ABS_SCHED 1 // k_score_mfma: the products of the next point group interleaved with the vector instructions of this one
#endif
#ifndef PL_ABS_EXP
#define PL_ABS_EXP 0 // experiment builds of k_score_mfma (scripts/exp): 1 = no exact pass, 2 = no expansion either
#endif
#ifndef PL_ABS_WAVES
#define PL_ABS_WAVES 4 // wavefronts per SIMD k_score_mfma's register allocation aims at (48 accumulators + 16 bit fields + operands)
#endif
constexpr int kMfmaThreads = PL_MFMA_THREADS; // 8 wavefronts share one chunk of correspondences (LDS: 20 KB shared + 2.8 KB per wave)

// Workgroup -> (hypothesis slice, chunk of correspondences) of the matrix-core scorers.  The operands of a hypothesis slice are
// streamed by the workgroups of ALL chunks of correspondences, and the hardware deals workgroups to the 8 XCDs - each with an L2
// of its own - round robin in linear order: with slice = blockIdx.x, chunk = blockIdx.y a slice's operands went through every
// L2, once per chunk (k_score_mfma2<2, 12>, 10 000 correspondences: 27 x 18 MB, FETCH_SIZE 285 MB per launch).  Here the
// (slice, chunk) pairs are numbered slice-major and XCD x (= linear workgroup index mod 8) takes the x-th eighth of them: an XCD
// sees 2 - 3 slices per launch (1 MB each, resident in its 4 MB L2 while their chunks' workgroups stream them).  Measured on one
// box, same build otherwise (PL_XCD_MAP = 0 / 1): FETCH_SIZE of the 7-point scorer 285 -> 61 MB per launch, of k_score_mfma<10>
// 20.9 -> 16.4 MB; grouped throughput 7-point 3.18 -> 3.41e8, P3P 7.38 -> 7.56e8, 5-point 1.61 -> 1.64e8 hypotheses/s,
// homography unchanged; a single problem's launch (all workgroups resident at once, not bandwidth bound) is unchanged.
__device__ __forceinline__ bool slice_chunk_of_workgroup(uint32_t slices, uint32_t chunks, uint32_t &slice, uint32_t &chunk) {
#if PL_XCD_MAP == 0
    slice = blockIdx.x, chunk = blockIdx.y;
    return slice < slices && chunk < chunks;
#else
    const uint32_t T = slices * chunks, L = blockIdx.x + gridDim.x * blockIdx.y;
    if (L >= T)
        return false;
    const uint32_t x = L & 7u, p = x * (T >> 3) + min(x, T & 7u) + (L >> 3);
    slice = p / chunks;
    chunk = p - slice * chunks;
    return true;
#endif
}

// The exact pass of k_score_mfma and k_score_mfmah (k_score_queue's arithmetic).  A drain is a dependency chain - queue entry, the lane
// permutation that turns the hypothesis into its record's slot, the record's twelve doubles from L2, the evaluation, the run
// totals -, not work: 31 of the kernel's 132 us with four wavefronts per SIMD to hide it (profiles/r06_score_phases.md).  So
// the chain is split - `fetch` issues every load of a batch of 64 pairs, `finish` evaluates it - and the kernel drains TWO
// batches at a time (128 waiting pairs), the second batch's loads in flight under the first one's chain (132.3 -> 129.2 us).
// Inlined into each of the sixteen expansion rounds of a tile (110 KB of instructions): as a real call (one copy, 15 KB) the
// saves and restores around 34 call sites cost more than the instruction cache gives back (137.9 us); one drain site behind
// the rounds keeps the sixteen bit fields alive across it and spills them once per tile (97 registers).
// queue: the wave's ring (kMfmaQueueCap entries: hypothesis of the unit << 9 | correspondence of the chunk); n0 (+ n1) pairs wait
// at `first`; pts: the chunk's correspondences in LDS, [5 or 4][npw]; unit_slots: lane l = record slot of the unit's hypothesis l.
constexpr int kMfmaQueueCap = 1024; // >= 127 waiting + 64 lanes * 10 point groups appended by one round
template <int EST>
__device__ __forceinline__ void mfma_drain(const uint16_t *queue, uint32_t first, uint32_t n0, uint32_t n1, const double *pts, int npw,
                                       const double *__restrict__ models, uint32_t unit_slots, double thr2, double *acc_s,
                                       uint32_t *acc_c) {
    const int lane = threadIdx.x & 63;
    struct Batch {
        uint32_t g;
        bool act;
        double x[EST == EST_ABS ? 5 : 4];
        double M[kModelDoubles];
    };
    auto fetch = [&](uint32_t n, uint32_t at, Batch &b) {
        b.act = (uint32_t)lane < n;
        const uint32_t e = b.act ? (uint32_t)queue[(at + lane) & (kMfmaQueueCap - 1)] : 0xffffu;
        const uint32_t pi = b.act ? (e & 0x1ffu) : 0u;
        b.g = e >> 9;
#pragma unroll
        for (int d = 0; d < (EST == EST_ABS ? 5 : 4); ++d)
            b.x[d] = pts[d * npw + pi];
        // the fp64 model straight from its record (hypothesis k lives in slot slots[k]): no hypothesis-ordered copy of the
        // models is needed on this path.  The unit's slot numbers were fetched when the unit began - one lane per
        // hypothesis -, so the record's address costs a lane permutation here, not a second dependent trip to memory.
        const uint32_t slot_g = (uint32_t)__shfl((int)unit_slots, (int)(b.act ? b.g : 0u), 64);
#if PL_ABS_EXP == 4 // (experiment: every pair against the unit's first model - no scattered loads; timing only)
        const double *Mp = models + (size_t)__builtin_amdgcn_readfirstlane((int)unit_slots) * kModelStride + (slot_g & 0u);
#else
        const double *Mp = models + (size_t)slot_g * kModelStride;
#endif
#pragma unroll
        for (int i = 0; i < kModelDoubles; ++i)
            b.M[i] = Mp[i];
    };
    auto finish = [&](const Batch &b) {
        double r2;
        const bool in = eval_point<EST>(b.M, b.x, thr2, r2) && b.act;
        const double v = in ? r2 : 0.0;
        const uint64_t inmask = __builtin_amdgcn_ballot_w64(in);
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.g);
        if (inmask && !__builtin_amdgcn_ballot_w64(b.act && b.g != g0)) { // one hypothesis: plain wave sum (k_score_queue)
            const double tot = wave_sum_dpp(v);
            if (lane == 0) {
                acc_s[g0] += tot;
                acc_c[g0] += (uint32_t)__popcll(inmask);
            }
        } else if (inmask) {
            add_run_totals(v, inmask, b.g, b.act, lane, acc_s, acc_c);
        }
    };
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    Batch b0, b1;
    fetch(n0, first, b0);
    if (n1) { // wave-uniform
        fetch(n1, first + n0, b1);
        finish(b0);
        finish(b1);
    } else {
        finish(b0);
    }
}

template <int PG>
__device__ __forceinline__ void score_mfma_body(const PointSet &pts, const uint4 *__restrict__ shadow16,
                                                const double *__restrict__ models, const uint32_t *__restrict__ slots,
                                                const uint32_t *__restrict__ num_hyp_ptr, uint32_t hyp_capacity,
                                                double thr2, const PrefilterArgs &pf, uint32_t *__restrict__ part_count,
                                                double *__restrict__ part_score, uint32_t slice, uint32_t chunk,
                                                uint32_t nslices) {
    constexpr int kWaves = kMfmaThreads / 64;
    constexpr int NPW = 32 * PG; // correspondences per chunk
    __shared__ double s_pts[5][NPW];
    __shared__ uint16_t s_queue[kWaves][kMfmaQueueCap]; // entries: hypothesis slot << 9 | correspondence of the chunk
    __shared__ double s_acc_s[kWaves][64];
    __shared__ uint32_t s_acc_c[kWaves][64];
    __shared__ uint32_t s_next_unit;
    __shared__ uint4 s_b[1 + kAbs16Dirs][PG][32]; // B operands: [0] first k block (X_hi, X_lo, 1, w), shared by the three
                                                  // instructions; [1 + d] second k block of direction d: (p X)_hi, (p X)_lo, p, |p|
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 31, half = lane >> 5;

    // ---- stationary operand: PG groups of 32 correspondences; lane l carries column l % 32 of every group ----
    // The per-point data of the PG tiles live in LDS, not in registers: the tile loop below is a real loop (an
    // unrolled one makes the register allocator give every tile its own 16 accumulators and spill the rest).
    uint32_t validbits = 0; // bit (PG - 1 - g): group g holds a real correspondence in this column
#pragma unroll
    for (int g = 0; g < PG; ++g)
        validbits |= (chunk * NPW + g * 32 + col < pts.n) ? (1u << (PG - 1 - g)) : 0u;
    // one thread per correspondence of the chunk: fp64 copy for the exact pass, fp16 operands for the filter
    // (pl_prefilter.h pf16_abs_point: the host test build runs the same function)
    for (uint32_t j = threadIdx.x; j < (uint32_t)NPW; j += kMfmaThreads) {
        const uint32_t i = chunk * NPW + j;
        const bool valid = i < pts.n;
        const uint32_t ic = valid ? i : 0u;
        double x[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            x[d] = pts.a[d][ic];
            s_pts[d][j] = x[d];
        }
        Abs16Point o;
        pf16_abs_point(x[0], x[1], x[2], x[3], x[4], valid, pf.g16, pf.thr, o);
        auto row = [](const uint16_t *h) {
            return make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                              (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
        };
        s_b[0][j >> 5][j & 31] = row(o.b0);
#pragma unroll
        for (int d = 0; d < kAbs16Dirs; ++d)
            s_b[1 + d][j >> 5][j & 31] = row(o.bp[d]);
    }
    if (threadIdx.x == 0)
        s_next_unit = 0;
    __syncthreads(); // the only workgroup barrier

    const uint32_t H = *as_uniform(num_hyp_ptr);
    uint16_t *const queue = s_queue[wave];
    double *const acc_s = s_acc_s[wave];
    uint32_t *const acc_c = s_acc_c[wave];
    const uint32_t waves_per_chunk = nslices * kWaves;
    auto request_ticket = [&]() -> uint32_t { // per-lane value; lane 0 holds the workgroup's next unit
        uint32_t t = 0;
        if (lane == 0) {
            const uint32_t k = atomicAdd(&s_next_unit, 1u);
            t = slice * kWaves + (k % kWaves) + (k / kWaves) * waves_per_chunk;
        }
        return t;
    };
    uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)request_ticket());
    uint32_t kb, gn;
    while (unit_of_ticket<32u>(ticket, H, waves_per_chunk, kb, gn)) {
        const uint32_t pending = request_ticket(); // the next unit's index travels while this one is evaluated
        const uint32_t unit_slots = slots[kb + min((uint32_t)lane, gn - 1u)]; // lane l: the record of the unit's hypothesis l
        acc_s[lane] = 0.0;
        acc_c[lane] = 0;
        uint32_t qhead = 0, qtail = 0;

        // The exact pass: mfma_drain above, two batches of 64 pairs at a time
        auto drain_pair = [&](uint32_t n2) { // 64 + n2 waiting pairs (0 < n2 <= 64)
#if PL_ABS_EXP != 1
            mfma_drain<EST_ABS>(queue, qhead, 64u, n2, &s_pts[0][0], NPW, models, unit_slots, thr2, acc_s, acc_c);
#endif
            qhead += 64u + n2;
        };
        auto drain = [&](uint32_t n) { // n <= 64 waiting pairs
#if PL_ABS_EXP != 1
            mfma_drain<EST_ABS>(queue, qhead, n, 0u, &s_pts[0][0], NPW, models, unit_slots, thr2, acc_s, acc_c);
#endif
            qhead += n;
        };

        const uint32_t ngroups32 = (gn + 31u) / 32u;
        // lane l: row l % 32 of the group; lanes 0..31 carry the first k block of the direction, lanes 32..63 the second one (the
        // same for the three instructions)
        auto load_a = [&](uint32_t hg, uint4 (&A)[kAbs16Dirs]) {
            const uint4 *row = shadow16 + (size_t)(kb + 32u * hg + (uint32_t)col) * 4;
#pragma unroll
            for (int d = 0; d < kAbs16Dirs; ++d)
                A[d] = row[half ? 3 : d];
        };
        uint4 Araw[kAbs16Dirs];
        load_a(0, Araw);
        for (uint32_t hg = 0; hg < ngroups32; ++hg) {
            half8_t Aop[kAbs16Dirs];
#pragma unroll
            for (int d = 0; d < kAbs16Dirs; ++d)
                __builtin_memcpy(&Aop[d], &Araw[d], 16);
            if (hg + 1 < ngroups32) // next group's operands travel while this one is evaluated
                load_a(hg + 1, Araw);
            uint32_t out[16]; // register v: bit (PG - 1 - g) = point group g is a proven outlier of hypothesis row(v)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                out[v] = 0u;
            const float16_t kZero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            // The loop over the point groups, software-pipelined by hand: the three products of group g + 1 are issued BETWEEN the
            // v_alignbit of group g (whose OR of sign bits sits in T), five or six vector instructions apart - an MFMA that has to
            // wait for the matrix pipe (32 cycles per product) blocks the SIMD's vector issue for every wavefront
            // (scripts/exp/overlap.cc), and three products back to back did that for 48 cycles per group.
            auto products = [&](int g, float16_t &D0, float16_t &D1, float16_t &D2) {
                half8_t Bop[kAbs16Dirs];
#pragma unroll
                for (int d = 0; d < kAbs16Dirs; ++d) {
                    const uint4 braw = s_b[half ? 1 + d : 0][g][col];
                    __builtin_memcpy(&Bop[d], &braw, 16);
                }
#if PL_ABS_EXP == 5 // (experiment: no matrix products - timing only, with PL_ABS_EXP == 2's missing expansion)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    D0[v] = (float)Bop[0][v & 7] + (float)Aop[0][v & 7];
                    D1[v] = (float)Bop[1][v & 7] + (float)Aop[1][v & 7];
                    D2[v] = (float)Bop[2][v & 7] + (float)Aop[2][v & 7];
                }
#else
                D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[0], Bop[0], kZero, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[1], Bop[1], kZero, 0, 0, 0);
                D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[2], Bop[2], kZero, 0, 0, 0);
#endif
            };
            float16_t D0, D1, D2;
            products(0, D0, D1, D2);
#pragma unroll PL_ABS_TILE_UNROLL
            for (int g = 1; g < PG; ++g) {
                uint32_t T[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) // proven outlier <=> one of the three signed distances is negative: OR of the sign bits (v_or3)
                    T[v] = __float_as_uint(D0[v]) | __float_as_uint(D1[v]) | __float_as_uint(D2[v]);
                products(g, D0, D1, D2);
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    out[v] = __builtin_amdgcn_alignbit(out[v], T[v], 31);
#if PL_ABS_SCHED
#if PL_ABS_SCHED == 1
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0); // the v_or3
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // the operands of the next group from LDS
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#else
                // the four v_or3 that free the registers the operands are loaded into, the loads, the other v_or3 under the loads' latency
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#endif
#endif
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const uint32_t sg = __float_as_uint(D0[v]) | __float_as_uint(D1[v]) | __float_as_uint(D2[v]);
                out[v] = __builtin_amdgcn_alignbit(out[v], sg, 31);
            }
            // ---- expansion: register v = hypothesis rows 8 (v / 4) + v % 4 (lanes 0..31) and + 4 (lanes 32..63) ----
            // (only the last, partial group of a short unit has slots without a hypothesis: the per-lane test "slot < gn" - three
            // vector instructions per round - runs behind a SCALAR branch for that group only; round 5)
            const bool partial_group = hg * 32u + 32u > gn;
            auto expand_round = [&](int v, bool check_slot) {
#if PL_ABS_EXP == 2 || PL_ABS_EXP == 5 // (experiment: no expansion, no exact pass - timing only, results are wrong)
                if (out[v] == 0x12345u)
                    qtail += 1;
                return;
#endif
                const uint32_t slot = hg * 32u + 8u * (uint32_t)(v >> 2) + 4u * (uint32_t)half + (uint32_t)(v & 3); // hypothesis index inside the unit
                uint32_t bits = ~out[v] & validbits;
                if (check_slot && slot >= gn)
                    bits = 0u;
                const uint64_t anyb = __builtin_amdgcn_ballot_w64(bits != 0u);
                if (anyb) { // wave-uniform
                    const uint32_t cnt = (uint32_t)__popc(bits);
                    if (!__builtin_amdgcn_ballot_w64(cnt > 1u)) {
                        // the usual case (1 % of the pairs survive: 0.1 bits per lane): at most ONE survivor per lane - its queue
                        // position is the number of lower lanes with a survivor (v_mbcnt, 2 instructions) instead of a DPP prefix
                        // sum over the lanes (12), and the bit loop is straight-line.  Same positions as the general path.
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(anyb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)anyb, 0u));
                        if (bits) {
                            const uint32_t g = (uint32_t)(PG - 1 - (31 - __clz((int)bits)));
                            queue[(qtail + below) & (kMfmaQueueCap - 1)] = (uint16_t)((slot << 9) | (g * 32u + (uint32_t)col));
                        }
                        qtail += (uint32_t)__popcll(anyb);
                    } else {
                        const uint32_t incl = wave_scan_u32(cnt); // inclusive prefix (lanes 0..31 = their hypothesis first)
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        uint32_t pos = qtail + incl - cnt;
                        uint32_t rest = bits;
                        while (rest) { // point groups in ascending order = bits from the top
                            const int hi = 31 - __clz((int)rest);
                            rest &= ~(1u << hi);
                            const uint32_t g = (uint32_t)(PG - 1 - hi);
                            queue[pos & (kMfmaQueueCap - 1)] = (uint16_t)((slot << 9) | (g * 32u + (uint32_t)col));
                            ++pos;
                        }
                        qtail += total;
                    }
                    while (qtail - qhead >= 128u)
                        drain_pair(64u);
                }
            };
            if (partial_group) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    expand_round(v, true);
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    expand_round(v, false);
            }
        }
        if (qtail - qhead > 64u)
            drain_pair(qtail - qhead - 64u);
        else if (qtail != qhead)
            drain(qtail - qhead);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((uint32_t)lane < gn) {
            const size_t o = (size_t)chunk * hyp_capacity + kb + lane;
            part_score[o] = acc_s[lane];
            part_count[o] = acc_c[lane];
        }
        ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)pending);
    }
}

template <int PG>
__global__ __launch_bounds__(kMfmaThreads) __attribute__((amdgpu_waves_per_eu(PL_ABS_WAVES, 8))) void k_score_mfma(PointSet pts, const uint4 *__restrict__ shadow16,
                                                               const double *__restrict__ models,
                                                               const uint32_t *__restrict__ slots,
                                                               const uint32_t *__restrict__ num_hyp_ptr,
                                                               uint32_t hyp_capacity, double thr2, PrefilterArgs pf,
                                                               uint32_t *__restrict__ part_count,
                                                               double *__restrict__ part_score,
                                                               uint32_t *__restrict__ tickets) {
    (void)tickets;
    uint32_t slice, chunk;
    slice_chunk_of_workgroup(gridDim.x, gridDim.y, slice, chunk);
    score_mfma_body<PG>(pts, shadow16, models, slots, num_hyp_ptr, hyp_capacity, thr2, pf, part_count, part_score, slice, chunk,
                        gridDim.x);
}
template <int PG>
__global__ __launch_bounds__(kMfmaThreads) __attribute__((amdgpu_waves_per_eu(PL_ABS_WAVES, 8))) void k_score_mfma_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    uint32_t slice, chunk;
    if (!g.active || !g.use_mfma || !slice_chunk_of_workgroup(g.slices, g.chunks, slice, chunk))
        return;
    const ScoreArgs &a = g.score;
    score_mfma_body<PG>(a.pts, static_cast<const uint4 *>(a.shadow16), a.models, a.slots, a.num_hyp, a.hyp_capacity, a.thr2,
                        a.pf, a.part_count, a.part_score, slice, chunk, g.slices);
}

// ---- two-view Sampson scores: the pre-filter on the matrix cores ----------------------------------------------------
// k_score_mfma2<EST, PG> (EST = relative pose / fundamental matrix): same contract, same queue and the same exact pass as
// k_score_queue, but pass A is three v_mfma_f32_32x32x16_f16 per 32 hypotheses x 32 correspondences: two accumulate the
// bilinear form C~ = b^T F a over fp16 high / low splits of both factors, one the quadratic forms S~ = Cx + Cy plus the
// slack (operands and error bounds: pl_prefilter.h "Sampson, fp16 / MFMA form"; hypothesis side built by k_sampson16,
// correspondence side here, once per workgroup, into LDS).  What is left for the vector ALU per pair is one
// multiplication and one FMA whose sign bit is the verdict (19 packed + 2 compares per PAIR of pairs in the fp32 filter).
// Accumulator layout (both tiles): lane l = correspondence l % 32 of the group, register v = hypothesis row
// 8 (v / 4) + 4 (l / 32) + v % 4.  The sign bits are shifted into one bit field per register over the PG groups of the
// chunk, then expanded into the wave's LDS queue register by register - lanes 0..31 (one hypothesis) before lanes 32..63
// (another), so every hypothesis is one run of the queue, which is all the drain's segmented sum needs.

template <int EST, int PG>
__device__ __forceinline__ void score_mfma2_body(const PointSet &pts, const uint4 *__restrict__ hypop,
                                                 const double *__restrict__ models, const uint32_t *__restrict__ slots,
                                                 const uint32_t *__restrict__ num_hyp_ptr, uint32_t hyp_capacity,
                                                 double thr2, const PrefilterArgs &pf, uint32_t *__restrict__ part_count,
                                                 double *__restrict__ part_score, uint32_t slice, uint32_t chunk,
                                                 uint32_t nslices) {
    static_assert(EST == EST_REL || EST == EST_FUND, "Sampson scores");
    constexpr int kWaves = kMfmaThreads / 64;
    constexpr int NPW = 32 * PG; // correspondences per chunk
    constexpr int NBR = (EST == EST_REL) ? 6 : 1;
    __shared__ double s_pts[4][NPW];
    __shared__ double s_bear[NBR][(EST == EST_REL) ? NPW : 1];
    __shared__ uint16_t s_queue[kWaves][kMfmaQueueCap]; // entries: hypothesis slot << 9 | correspondence of the chunk
    __shared__ double s_acc_s[kWaves][64];
    __shared__ uint32_t s_acc_c[kWaves][64];
    __shared__ uint32_t s_next_unit;
    __shared__ uint4 s_bop[PG][3][64]; // B operands: group, instruction, lane (column l % 32, k block l / 32)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 31, half = lane >> 5;

    // ---- stationary side: the chunk's correspondences (fp64 for the exact pass, fp16 operands for the filter) ----
    for (uint32_t j = threadIdx.x; j < (uint32_t)NPW; j += kMfmaThreads) {
        const uint32_t i = chunk * NPW + j;
        const uint32_t ic = i < pts.n ? i : 0u;
        const double a0 = pts.a[0][ic], a1 = pts.a[1][ic], b0 = pts.a[2][ic], b1 = pts.a[3][ic];
        s_pts[0][j] = a0, s_pts[1][j] = a1, s_pts[2][j] = b0, s_pts[3][j] = b1;
        if constexpr (EST == EST_REL) {
            const Vec3 u1 = bearing(a0, a1), u2 = bearing(b0, b1);
            s_bear[0][j] = u1.x, s_bear[1][j] = u1.y, s_bear[2][j] = u1.z;
            s_bear[3][j] = u2.x, s_bear[4][j] = u2.y, s_bear[5][j] = u2.z;
        }
    }
    uint32_t validbits = 0; // bit (PG - 1 - g): group g holds a real correspondence in this column
#pragma unroll
    for (int g = 0; g < PG; ++g)
        validbits |= (chunk * NPW + g * 32 + col < pts.n) ? (1u << (PG - 1 - g)) : 0u;
    // one thread per correspondence: its fp16 operand blocks (block 2 j + h of the operand = instruction j, lane half h)
    for (uint32_t j = threadIdx.x; j < (uint32_t)NPW; j += kMfmaThreads) {
        const uint32_t i = chunk * NPW + j;
        const bool valid = i < pts.n;
        const uint32_t ic = valid ? i : 0u;
        Sampson16Operand o;
        pf16_sampson_point(pts.a[0][ic], pts.a[1][ic], pts.a[2][ic], pts.a[3][ic], valid, pf.t16, o);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const uint16_t *h = b < 4 ? o.c + 8 * b : o.s + 8 * (b - 4);
            s_bop[j >> 5][b >> 1][(j & 31) + 32 * (b & 1)] =
                make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                           (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
        }
    }
    if (threadIdx.x == 0)
        s_next_unit = 0;
    __syncthreads(); // the only workgroup barrier

    const uint32_t H = *as_uniform(num_hyp_ptr);
    const float t16 = pf.t16;
    uint16_t *const queue = s_queue[wave];
    double *const acc_s = s_acc_s[wave];
    uint32_t *const acc_c = s_acc_c[wave];
    const uint32_t waves_per_chunk = nslices * kWaves;
    auto request_ticket = [&]() -> uint32_t {
        uint32_t t = 0;
        if (lane == 0) {
            const uint32_t k = atomicAdd(&s_next_unit, 1u);
            t = slice * kWaves + (k % kWaves) + (k / kWaves) * waves_per_chunk;
        }
        return t;
    };
    uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)request_ticket());
    uint32_t kb, gn;
    while (unit_of_ticket(ticket, H, waves_per_chunk, kb, gn)) {
        const uint32_t pending = request_ticket();
        const uint32_t unit_slots = slots[kb + min((uint32_t)lane, gn - 1u)]; // lane l: the record of the unit's hypothesis l (k_score_mfma)
        acc_s[lane] = 0.0;
        acc_c[lane] = 0;
        uint32_t qhead = 0, qtail = 0;

        auto drain = [&](uint32_t n) { // k_score_queue's exact pass; the fp64 models straight from their records
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool act = (uint32_t)lane < n;
            const uint32_t e = act ? (uint32_t)queue[(qhead + lane) & (kMfmaQueueCap - 1)] : 0xffffu;
            const uint32_t g = e >> 9, pi = act ? (e & 0x1ffu) : 0u;
            const double x0 = s_pts[0][pi], x1 = s_pts[1][pi], x2 = s_pts[2][pi], x3 = s_pts[3][pi];
            const double *Mp = models + (size_t)(uint32_t)__shfl((int)unit_slots, (int)(act ? g : 0u), 64) * kModelStride;
            double M[kModelDoubles];
#pragma unroll
            for (int i = 0; i < kModelDoubles; ++i)
                M[i] = Mp[i];
            double r2 = sampson_sq(M + kMatOff, x0, x1, x2, x3);
            bool in = r2 < thr2;
            if constexpr (EST == EST_REL) {
                if (in) {
                    Quat q;
                    q.w = M[0], q.x = M[1], q.y = M[2], q.z = M[3];
                    in = check_cheirality(q, v3(M[4], M[5], M[6]), v3(s_bear[0][pi], s_bear[1][pi], s_bear[2][pi]),
                                          v3(s_bear[3][pi], s_bear[4][pi], s_bear[5][pi]), 0.01);
                }
            }
            in = in && act;
            double v = in ? r2 : 0.0;
            const uint64_t inmask = __builtin_amdgcn_ballot_w64(in);
            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
            if (inmask && !__builtin_amdgcn_ballot_w64(act && g != g0)) {
                const double tot = wave_sum_dpp(v);
                if (lane == 0) {
                    acc_s[g0] += tot;
                    acc_c[g0] += (uint32_t)__popcll(inmask);
                }
            } else if (inmask) {
                add_run_totals(v, inmask, g, act, lane, acc_s, acc_c);
            }
            qhead += n;
        };

        const uint32_t ngroups32 = (gn + 31u) / 32u;
        auto load_a = [&](uint32_t hg, uint4 (&A)[3]) {
            const uint4 *row = hypop + (size_t)(kb + 32u * hg + (uint32_t)col) * 6;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                A[j] = row[2 * j + half];
        };
        uint4 Araw[3];
        load_a(0, Araw);
        for (uint32_t hg = 0; hg < ngroups32; ++hg) {
            half8_t Aop[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_memcpy(&Aop[j], &Araw[j], 16);
            if (hg + 1 < ngroups32) // next group's operands travel while this one is evaluated
                load_a(hg + 1, Araw);
            uint32_t out[16]; // register v: bit (PG - 1 - g) = point group g is a proven outlier of hypothesis row(v)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                out[v] = 0u;
            const float16_t kZero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
            for (int g = 0; g < PG; ++g) {
                half8_t Bop[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const uint4 braw = s_bop[g][j][lane];
                    __builtin_memcpy(&Bop[j], &braw, 16);
                }
                float16_t C = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[0], Bop[0], kZero, 0, 0, 0);
                C = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[1], Bop[1], C, 0, 0, 0);
                const float16_t S = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[2], Bop[2], kZero, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    // proven outlier <=> C^2 > t16 S' <=> t16 S' - C^2 < 0: the sign bit goes into the bit field
                    const float d = fmaf(t16, S[v], -(C[v] * C[v]));
                    out[v] = __builtin_amdgcn_alignbit(out[v], __float_as_uint(d), 31);
                }
            }
            // ---- expansion: register v = hypothesis rows 8 (v / 4) + v % 4 (lanes 0..31) and + 4 (lanes 32..63) ----
            const bool partial_group = hg * 32u + 32u > gn; // (the slot test behind a scalar branch: k_score_mfma's expansion says why)
            auto expand_round = [&](int v, bool check_slot) {
                const uint32_t slot = hg * 32u + 8u * (uint32_t)(v >> 2) + 4u * (uint32_t)half + (uint32_t)(v & 3);
                uint32_t bits = ~out[v] & validbits;
                if (check_slot && slot >= gn)
                    bits = 0u;
                const uint64_t anyb = __builtin_amdgcn_ballot_w64(bits != 0u);
                if (anyb) { // wave-uniform
                    const uint32_t cnt = (uint32_t)__popc(bits);
                    if (!__builtin_amdgcn_ballot_w64(cnt > 1u)) {
                        // the usual case (1 % of the pairs survive: 0.1 bits per lane): at most ONE survivor per lane - its queue
                        // position is the number of lower lanes with a survivor (v_mbcnt, 2 instructions) instead of a DPP prefix
                        // sum over the lanes (12), and the bit loop is straight-line.  Same positions as the general path.
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(anyb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)anyb, 0u));
                        if (bits) {
                            const uint32_t g = (uint32_t)(PG - 1 - (31 - __clz((int)bits)));
                            queue[(qtail + below) & (kMfmaQueueCap - 1)] = (uint16_t)((slot << 9) | (g * 32u + (uint32_t)col));
                        }
                        qtail += (uint32_t)__popcll(anyb);
                    } else {
                        const uint32_t incl = wave_scan_u32(cnt); // inclusive prefix (lanes 0..31 = their hypothesis first)
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        uint32_t pos = qtail + incl - cnt;
                        uint32_t rest = bits;
                        while (rest) { // point groups in ascending order = bits from the top
                            const int hi = 31 - __clz((int)rest);
                            rest &= ~(1u << hi);
                            const uint32_t g = (uint32_t)(PG - 1 - hi);
                            queue[pos & (kMfmaQueueCap - 1)] = (uint16_t)((slot << 9) | (g * 32u + (uint32_t)col));
                            ++pos;
                        }
                        qtail += total;
                    }
                    while (qtail - qhead >= 64u)
                        drain(64u);
                }
            };
            if (partial_group) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    expand_round(v, true);
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    expand_round(v, false);
            }
        }
        while (qtail != qhead)
            drain(min(64u, qtail - qhead));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((uint32_t)lane < gn) {
            const size_t o = (size_t)chunk * hyp_capacity + kb + lane;
            part_score[o] = acc_s[lane];
            part_count[o] = acc_c[lane];
        }
        ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)pending);
    }
}

template <int EST, int PG>
__global__ __launch_bounds__(kMfmaThreads) void k_score_mfma2(PointSet pts, const uint4 *__restrict__ hypop,
                                                              const double *__restrict__ models,
                                                              const uint32_t *__restrict__ slots,
                                                              const uint32_t *__restrict__ num_hyp_ptr,
                                                              uint32_t hyp_capacity, double thr2, PrefilterArgs pf,
                                                              uint32_t *__restrict__ part_count,
                                                              double *__restrict__ part_score) {
    uint32_t slice, chunk;
    slice_chunk_of_workgroup(gridDim.x, gridDim.y, slice, chunk);
    score_mfma2_body<EST, PG>(pts, hypop, models, slots, num_hyp_ptr, hyp_capacity, thr2, pf, part_count, part_score, slice,
                              chunk, gridDim.x);
}

template <int EST, int PG> __global__ __launch_bounds__(kMfmaThreads) void k_score_mfma2_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    uint32_t slice, chunk;
    if (!g.active || !g.use_mfma || !slice_chunk_of_workgroup(g.slices, g.chunks, slice, chunk))
        return;
    const ScoreArgs &a = g.score;
    score_mfma2_body<EST, PG>(a.pts, static_cast<const uint4 *>(a.shadow16), a.models, a.slots, a.num_hyp, a.hyp_capacity,
                              a.thr2, a.pf, a.part_count, a.part_score, slice, chunk, g.slices);
}

// ---- homography: the pre-filter on the matrix cores (round 3) ------------------------------------------------------------
// k_score_mfmah<PG>: same contract, same queue and the same exact pass as k_score_queue<EST_HOM>, pass A out of the matrix
// pipe: TWO chained v_mfma_f32_32x32x16_f16 (32 k slots: fp16 high / low splits of both factors) per 8 hypotheses x 32
// correspondences deliver, for every pair, the four linear forms V_0 = h_0 - b_0 h_2, V_1 = h_1 - b_1 h_2, U = thr h_2 and
// the slack S (rows 4 j + r of hypothesis j; operands and error budget: pl_prefilter.h "homography, fp16 / MFMA form";
// hypothesis side built by k_hom16, correspondence side here, once per workgroup, into LDS).  The sign of h_2 is not fixed over an image, so the inlier region is a double cone and the
// verdict is  max(|V_0|, |V_1|) > |U| + S:  four vector instructions per pair (v_max with |.| modifiers, v_add, v_sub,
// v_alignbit into the per-lane bit field), against 84 per (hypothesis, 320 correspondences) of the fp32 form.
// Accumulator layout: lane l = correspondence l % 32 of the group, register 4 q + r = row r of hypothesis 2 q + l / 32.
template <int PG>
__device__ __forceinline__ void score_mfmah_body(const PointSet &pts, const uint4 *__restrict__ hypop,
                                                 const double *__restrict__ models, const uint32_t *__restrict__ slots,
                                                 const uint32_t *__restrict__ num_hyp_ptr, uint32_t hyp_capacity,
                                                 double thr2, const PrefilterArgs &pf, uint32_t *__restrict__ part_count,
                                                 double *__restrict__ part_score, uint32_t slice, uint32_t chunk,
                                                 uint32_t nslices) {
    constexpr int kWaves = kMfmaThreads / 64;
    constexpr int NPW = 32 * PG; // correspondences per chunk
    __shared__ double s_pts[4][NPW];
    __shared__ uint16_t s_queue[kWaves][kMfmaQueueCap]; // entries: hypothesis slot << 9 | correspondence of the chunk
    __shared__ double s_acc_s[kWaves][64];
    __shared__ uint32_t s_acc_c[kWaves][64];
    __shared__ uint32_t s_next_unit;
    __shared__ uint4 s_bop[PG][4][32]; // B operand: group, k block (8 slots each), column
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 31, half = lane >> 5;

    uint32_t validbits = 0; // bit (PG - 1 - g): group g holds a real correspondence in this column
#pragma unroll
    for (int g = 0; g < PG; ++g)
        validbits |= (chunk * NPW + g * 32 + col < pts.n) ? (1u << (PG - 1 - g)) : 0u;
    // one thread per correspondence of the chunk: fp64 copy for the exact pass, fp16 operands for the filter
    for (uint32_t j = threadIdx.x; j < (uint32_t)NPW; j += kMfmaThreads) {
        const uint32_t i = chunk * NPW + j;
        const bool valid = i < pts.n;
        const uint32_t ic = valid ? i : 0u;
        const double a0 = pts.a[0][ic], a1 = pts.a[1][ic], b0 = pts.a[2][ic], b1 = pts.a[3][ic];
        s_pts[0][j] = a0, s_pts[1][j] = a1, s_pts[2][j] = b0, s_pts[3][j] = b1;
        Hom16Point o;
        pf16_hom_point(a0, a1, b0, b1, valid, pf.h16, o);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint16_t *h = o.k + 8 * b;
            s_bop[j >> 5][b][j & 31] = make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                                                  (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
        }
    }
    if (threadIdx.x == 0)
        s_next_unit = 0;
    __syncthreads(); // the only workgroup barrier

    const uint32_t H = *as_uniform(num_hyp_ptr);
    uint16_t *const queue = s_queue[wave];
    double *const acc_s = s_acc_s[wave];
    uint32_t *const acc_c = s_acc_c[wave];
    const uint32_t waves_per_chunk = nslices * kWaves;
    auto request_ticket = [&]() -> uint32_t { // per-lane value; lane 0 holds the workgroup's next unit
        uint32_t t = 0;
        if (lane == 0) {
            const uint32_t k = atomicAdd(&s_next_unit, 1u);
            t = slice * kWaves + (k % kWaves) + (k / kWaves) * waves_per_chunk;
        }
        return t;
    };
    uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)request_ticket());
    uint32_t kb, gn;
    while (unit_of_ticket(ticket, H, waves_per_chunk, kb, gn)) {
        const uint32_t pending = request_ticket(); // the next unit's index travels while this one is evaluated
        const uint32_t unit_slots = slots[kb + min((uint32_t)lane, gn - 1u)]; // lane l: the record of the unit's hypothesis l (k_score_mfma)
        acc_s[lane] = 0.0;
        acc_c[lane] = 0;
        uint32_t qhead = 0, qtail = 0;

        // the exact pass: mfma_drain (k_score_mfma's), ONE batch of 64 pairs at a time - two at a time need 16 registers more than
        // the 84 that three workgroups per CU leave a wavefront: spilt, hom_10000 1.50 -> 1.42e8 on one box (round 6)
        auto drain = [&](uint32_t n) { // n <= 64 waiting pairs
            mfma_drain<EST_HOM>(queue, qhead, n, 0u, &s_pts[0][0], NPW, models, unit_slots, thr2, acc_s, acc_c);
            qhead += n;
        };

        const uint32_t ngroups8 = (gn + 7u) / 8u;
        // lane l: row l % 32 of the group's k blocks l / 32 (first instruction) and 2 + l / 32 (second)
        auto load_a = [&](uint32_t hg, uint4 (&A)[2]) {
            const uint4 *grp = hypop + ((size_t)(kb >> 3) + hg) * 128;
            A[0] = grp[32 * half + col];
            A[1] = grp[64 + 32 * half + col];
        };
        uint4 Araw[2];
        load_a(0, Araw);
        for (uint32_t hg = 0; hg < ngroups8; ++hg) {
            half8_t Aop[2];
            __builtin_memcpy(&Aop[0], &Araw[0], 16);
            __builtin_memcpy(&Aop[1], &Araw[1], 16);
            if (hg + 1 < ngroups8) // next group's operands travel while this one is evaluated
                load_a(hg + 1, Araw);
            uint32_t out[4] = {0u, 0u, 0u, 0u}; // hypothesis 2 q + half: bit (PG - 1 - g) = point group g is a proven outlier
            const float16_t kZero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
            for (int g = 0; g < PG; ++g) {
                const uint4 b0raw = s_bop[g][half][col], b1raw = s_bop[g][2 + half][col];
                half8_t Bop0, Bop1;
                __builtin_memcpy(&Bop0, &b0raw, 16);
                __builtin_memcpy(&Bop1, &b1raw, 16);
                float16_t D = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[0], Bop0, kZero, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aop[1], Bop1, D, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // proven outlier <=> max(|V_0|, |V_1|) > |U| + S: the sign bit of the difference goes into the bit field
                    // (v_med3_f32 |a|, |b|, 3e38 = the larger magnitude in ONE instruction - the forms stay below 2^20; fmaxf,
                    // or a median with +inf that the compiler folds into it, canonicalises both inputs first: two more
                    // instructions per pair.  No NaNs here: NaN models carry zero rows)
                    const float t = __builtin_amdgcn_fmed3f(__builtin_fabsf(D[4 * q]), __builtin_fabsf(D[4 * q + 1]), 3.0e38f);
                    const float d = (__builtin_fabsf(D[4 * q + 2]) + D[4 * q + 3]) - t;
                    out[q] = __builtin_amdgcn_alignbit(out[q], __float_as_uint(d), 31);
                }
            }
            // ---- expansion: four rounds, round q = hypothesis 2 q (lanes 0..31) and 2 q + 1 (lanes 32..63) ----
            const bool partial_group = hg * 8u + 8u > gn; // (the slot test behind a scalar branch: k_score_mfma's expansion says why)
            auto expand_round = [&](int q, bool check_slot) {
                const uint32_t slot = hg * 8u + 2u * (uint32_t)q + (uint32_t)half; // hypothesis index inside the unit
                uint32_t bits = ~out[q] & validbits;
                if (check_slot && slot >= gn)
                    bits = 0u;
                const uint64_t anyb = __builtin_amdgcn_ballot_w64(bits != 0u);
                if (anyb) { // wave-uniform
                    const uint32_t cnt = (uint32_t)__popc(bits);
                    if (!__builtin_amdgcn_ballot_w64(cnt > 1u)) {
                        // the usual case (1 % of the pairs survive: 0.1 bits per lane): at most ONE survivor per lane - its queue
                        // position is the number of lower lanes with a survivor (v_mbcnt, 2 instructions) instead of a DPP prefix
                        // sum over the lanes (12), and the bit loop is straight-line.  Same positions as the general path.
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(anyb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)anyb, 0u));
                        if (bits) {
                            const uint32_t g = (uint32_t)(PG - 1 - (31 - __clz((int)bits)));
                            queue[(qtail + below) & (kMfmaQueueCap - 1)] = (uint16_t)((slot << 9) | (g * 32u + (uint32_t)col));
                        }
                        qtail += (uint32_t)__popcll(anyb);
                    } else {
                        const uint32_t incl = wave_scan_u32(cnt); // inclusive prefix (lanes 0..31 = their hypothesis first)
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        uint32_t pos = qtail + incl - cnt;
                        uint32_t rest = bits;
                        while (rest) { // point groups in ascending order = bits from the top
                            const int hi = 31 - __clz((int)rest);
                            rest &= ~(1u << hi);
                            const uint32_t g = (uint32_t)(PG - 1 - hi);
                            queue[pos & (kMfmaQueueCap - 1)] = (uint16_t)((slot << 9) | (g * 32u + (uint32_t)col));
                            ++pos;
                        }
                        qtail += total;
                    }
                    while (qtail - qhead >= 64u)
                        drain(64u);
                }
            };
            if (partial_group) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    expand_round(q, true);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    expand_round(q, false);
            }
        }
        while (qtail != qhead)
            drain(min(64u, qtail - qhead));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((uint32_t)lane < gn) {
            const size_t o = (size_t)chunk * hyp_capacity + kb + lane;
            part_score[o] = acc_s[lane];
            part_count[o] = acc_c[lane];
        }
        ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)pending);
    }
}
template <int PG>
__global__ __launch_bounds__(kMfmaThreads) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_score_mfmah(
    PointSet pts, const uint4 *__restrict__ hypop, const double *__restrict__ models, const uint32_t *__restrict__ slots,
    const uint32_t *__restrict__ num_hyp_ptr, uint32_t hyp_capacity, double thr2, PrefilterArgs pf,
    uint32_t *__restrict__ part_count, double *__restrict__ part_score) {
    uint32_t slice, chunk;
    slice_chunk_of_workgroup(gridDim.x, gridDim.y, slice, chunk);
    score_mfmah_body<PG>(pts, hypop, models, slots, num_hyp_ptr, hyp_capacity, thr2, pf, part_count, part_score, slice, chunk,
                         gridDim.x);
}
template <int PG>
__global__ __launch_bounds__(kMfmaThreads) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_score_mfmah_g(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    uint32_t slice, chunk;
    if (!g.active || !g.use_mfma || !slice_chunk_of_workgroup(g.slices, g.chunks, slice, chunk))
        return;
    const ScoreArgs &a = g.score;
    score_mfmah_body<PG>(a.pts, static_cast<const uint4 *>(a.shadow16), a.models, a.slots, a.num_hyp, a.hyp_capacity, a.thr2,
                         a.pf, a.part_count, a.part_score, slice, chunk, g.slices);
}

// ---- MSAC score in the reference's summation order ------------------------------------------------------------------
// The streaming scorers add the inlier residuals of a hypothesis in tree order; the reference adds them one after the
// other in correspondence order (utils.cc:52-63).  The two sums differ in the last bits, which only matters when two
// hypotheses tie - but then it decides `score < best` (ransac_impl.h:114-116, 142-146).  So every score a decision
// is taken on - the candidates k_records lists (improving hypotheses plus anything within 1e-9 of the running
// minimum) and the re-scored refined models - is recomputed here exactly like the reference does it: one workgroup per
// model; the threads evaluate the correspondences of a chunk in parallel and compact the inliers' squared residuals
// in index order into LDS (block scan), one lane adds them sequentially.
// (32 KB of LDS: the workgroup fits next to two workgroups of the streaming scorers - with 64 KB it waited for a scoring
// launch's tail whenever another group's scorer held the device, like the orbit kernel's large build, pipeline.hip.
// 512 lanes: while one lane adds, the others only hold wave slots - with 1024 lanes two workgroups filled a CU's 32 slots and
// kept the scorers of other groups off it: batch_mixed 52.2 -> 55.6 k problems/s, hom_10000 +3 %; 256 lanes cost the
// single-problem path 6 % in the evaluation phase.  The order of the sum does not depend on the workgroup's shape.)
constexpr int kSeqThreads = 512, kSeqPerThread = 8, kSeqChunk = kSeqThreads * kSeqPerThread;

template <int EST> __device__ __forceinline__ void score_seq_body(const SeqScoreArgs &a) {
    constexpr int ND = point_doubles(EST);
    __shared__ __attribute__((aligned(16))) double s_list[kSeqChunk + 64];
    __shared__ uint32_t s_wave_tot[kSeqThreads / 64], s_total;
    __shared__ double s_sum;
    __shared__ uint32_t s_inliers;
    const uint32_t nrec = min(*a.num, a.cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.ctl_host && blockIdx.x == 0 && threadIdx.x == 0) {
        const BatchCtl c = *a.ctl_src; // final: every kernel that writes it is an earlier launch of the stream
        *a.ctl_host = c;
    }
    for (uint32_t r = blockIdx.x; r < nrec; r += gridDim.x) {
        const double *Mp = a.models + (size_t)(a.cand ? a.cand[r].slot : r) * kModelStride;
        double M[kModelDoubles];
#pragma unroll
        for (int i = 0; i < kModelDoubles; ++i)
            M[i] = Mp[i];
        if (threadIdx.x == 0) {
            s_sum = 0.0;
            s_inliers = 0;
        }
        uint32_t count = 0; // (thread 0 keeps the total)
        for (uint32_t base = 0; base < a.pts.n; base += kSeqChunk) {
            double r2v[kSeqPerThread];
            uint32_t flags = 0, cnt = 0, inl = 0; // cnt: terms this thread contributes to the list; inl: inliers among them
#pragma unroll
            for (int j = 0; j < kSeqPerThread; ++j) {
                const uint32_t i = base + threadIdx.x * kSeqPerThread + j; // contiguous per thread: index order
                r2v[j] = 0.0;
                if (i < a.pts.n) {
                    double x[ND];
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        x[d] = a.pts.a[d][i];
                    double r2;
                    const bool in = eval_point<EST>(M, x, a.thr2, r2);
                    if constexpr (EST == EST_ABS) {
                        if (in) { // utils.cc:57-60: the inliers' residuals, the outliers' share in one product at the end (:63)
                            r2v[j] = r2;
                            flags |= 1u << j;
                            ++cnt;
                        }
                    } else {
                        // utils.cc:188-198, 230-235, 320-325: the two-view scores add r^2 OR the squared threshold for
                        // every correspondence, in correspondence order - one term per correspondence in the list
                        r2v[j] = in ? r2 : a.thr2;
                        flags |= 1u << j;
                        ++cnt;
                        inl += in ? 1u : 0u;
                    }
                }
            }
            const uint32_t incl = wave_scan_u32(cnt);
            if (lane == 63)
                s_wave_tot[wave] = incl;
            __syncthreads(); // (also: the previous chunk's list has been consumed)
            uint32_t off = incl - cnt;
            for (int w = 0; w < wave; ++w)
                off += s_wave_tot[w];
#pragma unroll
            for (int j = 0; j < kSeqPerThread; ++j)
                if ((flags >> j) & 1u)
                    s_list[off++] = r2v[j];
            if (threadIdx.x == kSeqThreads - 1) {
                s_total = off;
                for (int z = 0; z < 64; ++z) // pad with +0.0 to the next multiple of 64 (x + 0.0 == x: the sum never is -0.0)
                    s_list[off + z] = 0.0;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t total = s_total;
                double sum = s_sum;
                // utils.cc:59 / :193 / :233 / :323, in correspondence order: a chain of dependent additions.  64 terms per inline-asm
                // statement (pl_lm_chain.inc: two terms per ds_read_b128, the reads six pairs ahead of the additions): 11.7 cycles
                // per term against 17 - 32 for the compiler's schedule of the same loop (scripts/exp/chain_add.cc)
                __builtin_amdgcn_s_setprio(3); // (a chain of dependent additions: first in line for the SIMD's issue port)
                for (uint32_t j = 0; j < total; j += 64) {
                    const uint32_t addr = (uint32_t)(uintptr_t)&s_list[j];
                    PL_LM_CHAIN64(sum, addr); // (+0.0 beyond `total`)
                }
                __builtin_amdgcn_s_setprio(0);
                s_sum = sum;
                if constexpr (EST == EST_ABS)
                    count += total;
            }
            if constexpr (EST != EST_ABS) { // inliers of the chunk (an integer: any order)
                const uint32_t wi = wave_sum_u32(inl);
                if (lane == 0 && wi)
                    atomicAdd(&s_inliers, wi);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if constexpr (EST != EST_ABS)
                count = s_inliers;
            const double score = (EST == EST_ABS) ? s_sum + (double)(a.pts.n - count) * a.thr2 /* utils.cc:63 */ : s_sum;
            if (a.cand) {
                a.cand[r].count = count;
                a.cand[r].score = score;
                if (r < a.host_cap) {
                    RecordMeta m = a.cand[r];
                    a.host_cand[r] = m;
                }
            } else {
                a.count[r] = count;
                a.score[r] = score;
                if (a.host_count) {
                    a.host_count[r] = count;
                    a.host_score[r] = score;
                }
            }
        }
        __syncthreads();
    }
}
template <int EST> __global__ __launch_bounds__(kSeqThreads) void k_score_seq(SeqScoreArgs a) { score_seq_body<EST>(a); }
template <int EST> __global__ __launch_bounds__(kSeqThreads) void k_score_seq_g(const SeqScoreArgs *arr) {
    const SeqScoreArgs &a = arr[blockIdx.z];
    if (a.cap == 0 || a.num == nullptr)
        return;
    score_seq_body<EST>(a);
}
// (the candidate re-scoring of a group's batch step: arguments inside the GroupArgs table)
template <int EST> __global__ __launch_bounds__(kSeqThreads) void k_score_seq_gb(const GroupArgs *ga) {
    const GroupArgs &g = ga[blockIdx.z];
    if (!g.active)
        return;
    score_seq_body<EST>(g.seq);
}

// ------------------------------------------------------------------------------------ mask
template <int EST>
__device__ __forceinline__ void mask_body(const PointSet &pts, const double *model, double thr2, uint8_t *mask, uint8_t *host_mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pts.n)
        return;
    double M[kModelDoubles];
    for (int k = 0; k < kModelDoubles; ++k)
        M[k] = model[k];
    bool in;
    if constexpr (EST == EST_ABS) {
        in = reproj_mask(M, pts.a[0][i], pts.a[1][i], pts.a[2][i], pts.a[3][i], pts.a[4][i], thr2);
    } else {
        double r2;
        const double pt[4] = {pts.a[0][i], pts.a[1][i], pts.a[2][i], pts.a[3][i]};
        in = eval_point<EST>(M, pt, thr2, r2);
    }
    mask[i] = in ? 1 : 0;
    if (host_mask)
        host_mask[i] = in ? 1 : 0;
}
template <int EST> __global__ __launch_bounds__(256) void k_mask(PointSet pts, const double *model, double thr2, uint8_t *mask,
                                                                 uint8_t *host_mask) {
    mask_body<EST>(pts, model, thr2, mask, host_mask);
}
template <int EST> __global__ __launch_bounds__(256) void k_mask_g(const MaskArgs *arr) {
    const MaskArgs &a = arr[blockIdx.z];
    if (blockIdx.x * 256u >= a.pts.n)
        return;
    mask_body<EST>(a.pts, a.model, a.thr2, a.mask, a.host_mask);
}

// ------------------------------------------------------------------------------------ LM
// starting point of a refinement task: its parameter block, or (tasks enqueued behind the kernel that chooses their model) the
// parameters of a model record in device memory - what driver.cc's params_from_record() extracts on the host
__device__ __forceinline__ void lm_start_params(int est, const LMTask &T, double *cur) {
    if (!T.start_record) {
        for (int i = 0; i < kParamDoubles; ++i)
            cur[i] = T.params[i];
        return;
    }
    for (int i = 0; i < kParamDoubles; ++i)
        cur[i] = 0.0;
    if (est == EST_ABS || est == EST_REL) {
        for (int i = 0; i < 7; ++i)
            cur[i] = T.start_record[i];
    } else {
        for (int i = 0; i < 9; ++i)
            cur[i] = T.start_record[kMatOff + i];
    }
}

template <int N> struct BlockReduce {
    // Reduces N per-thread doubles (+ one counter) over the 1024-thread workgroup.  Result in out[0..N)
    // and *count_out (valid for every thread after the call).  Fixed order: butterfly inside each
    // wavefront, then wavefronts 0..15 in sequence.
    __device__ static void run(const double *v, uint32_t cnt, double (*scratch)[N + 1], double *out,
                               uint32_t *count_out) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        // Round 5 (cycle counters, scripts/lm_profile.py): this was 43 % of an LM iteration of a small problem - 10.8 k cycles for
        // the 28 values of a pose, 16.2 k for the 45 of a homography: written value by value (reduce, then a lane-0 store) the
        // stores' EXEC changes kept the N dependent DPP chains (6 steps of two cross-lane moves and an addition, ~400 cycles
        // each) from overlapping.  Now step by step over all values - the same operations on every value, so the same bits -
        // with the N chains independent inside a step, and ONE lane-0 block of stores at the end.
        double s[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            s[i] = v[i];
#pragma unroll
        for (int i = 0; i < N; ++i)
            s[i] += dpp_move<0xB1>(s[i]); // quad_perm [1,0,3,2]
#pragma unroll
        for (int i = 0; i < N; ++i)
            s[i] += dpp_move<0x4E>(s[i]); // quad_perm [2,3,0,1]
#pragma unroll
        for (int i = 0; i < N; ++i)
            s[i] += dpp_move<0x141>(s[i]); // row_half_mirror
#pragma unroll
        for (int i = 0; i < N; ++i)
            s[i] += dpp_move<0x140>(s[i]); // row_mirror
#pragma unroll
        for (int i = 0; i < N; ++i)
            s[i] += dpp_move<0x142, 0xa>(s[i]); // row_bcast:15 into rows 1 and 3
#pragma unroll
        for (int i = 0; i < N; ++i)
            s[i] += dpp_move<0x143, 0xc>(s[i]); // row_bcast:31 into rows 2 and 3: lane 63 holds the total (wave_sum_dpp)
        const uint32_t c = wave_sum_u32(cnt);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                scratch[wave][i] = s[i];
        }
        if (lane == 0)
            scratch[wave][N] = (double)c;
        __syncthreads();
        if (threadIdx.x <= N) {
            double s = 0;
            for (int w = 0; w < kLMThreads / 64; ++w)
                s += scratch[w][threadIdx.x];
            if (threadIdx.x < N)
                out[threadIdx.x] = s;
            else
                *count_out = (uint32_t)s;
        }
        __syncthreads();
    }
};

// ---- the block reduction of k_lm, transposed (round 5) --------------------------------------------------------------------
// Cycle counters (scripts/lm_profile.py, profiles/r05_lm_profile.md): BlockReduce above was 43 % of an LM iteration of a small
// problem - 10.8 k cycles for the 28 sums of a pose, 16.2 k for the 46 of a homography - because every lane carries EVERY sum through
// all six butterfly steps (28 x 6 x 5 instructions), although only one lane's result is used.  Here a butterfly step HALVES the list
// a lane carries: lanes whose step bit is 0 keep the first half of the current list, lanes whose bit is 1 the second half, and each
// sends the other half to its partner (lane ^ 1, ^ 2, ^ 4, ^ 8 - quad_perm and banked row shifts): 28 -> 14 -> 7 -> 4 -> 2 values
// per lane after the four steps inside a row of 16 lanes, 27 exchanged pairs instead of 112.  The row sums go to LDS, and the
// thread that owns a sum adds its four rows as the butterfly's last two steps would - (R3 + R2) + (R1 + R0) - and then the
// wavefronts in sequence.  Every sum is built from the same partial sums in the same tree as before (additions commute bit for
// bit), so the results are the old bits; the counters take the old path.
template <int N> struct BlockReduceT {
    static constexpr int n1 = (N + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2; // list length after each step
    static constexpr int kStageDoubles = (kLMThreads / 64) * n4 * 64;
    // exchange of one double with the lane `lane ^ (1 << STEP)`
    template <int STEP> __device__ static __forceinline__ double xchg(double v) {
        const uint64_t b = (uint64_t)__double_as_longlong(v);
        int lo = (int)(uint32_t)b, hi = (int)(uint32_t)(b >> 32);
        if constexpr (STEP == 0) {
            lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
            hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, false);
        } else if constexpr (STEP == 1) {
            lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xf, 0xf, false); // quad_perm [2,3,0,1]
            hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xf, 0xf, false);
        } else if constexpr (STEP == 2) { // banks 0, 2 read four lanes up (row_shl:4), banks 1, 3 four lanes down (row_shr:4)
            int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xf, 0x5, false);
            l2 = __builtin_amdgcn_update_dpp(l2, lo, 0x114, 0xf, 0xa, false);
            int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xf, 0x5, false);
            h2 = __builtin_amdgcn_update_dpp(h2, hi, 0x114, 0xf, 0xa, false);
            lo = l2, hi = h2;
        } else { // banks 0, 1 read eight lanes up, banks 2, 3 eight lanes down
            int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x108, 0xf, 0x3, false);
            l2 = __builtin_amdgcn_update_dpp(l2, lo, 0x118, 0xf, 0xc, false);
            int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x108, 0xf, 0x3, false);
            h2 = __builtin_amdgcn_update_dpp(h2, hi, 0x118, 0xf, 0xc, false);
            lo = l2, hi = h2;
        }
        return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
    }
    template <int STEP, int NCUR> __device__ static __forceinline__ void halve(double *s, int lane) {
        constexpr int NN = (NCUR + 1) / 2;
        const bool upper = (lane >> STEP) & 1;
#pragma unroll
        for (int i = 0; i < NN; ++i) {
            const double a = s[i];
            const double b = (i + NN < NCUR) ? s[i + NN] : 0.0; // (a list of odd length: the missing partner is a zero nobody reads)
            const double keep = upper ? b : a, send = upper ? a : b;
            s[i] = keep + xchg<STEP>(send);
        }
    }
    // where sum v sits after the four steps: lane of the row (bits = the halves it went into), slot of that lane's list
    __device__ static __forceinline__ void home(int v, int &lane16, int &slot) {
        int idx = v, l = 0;
        if (idx >= n1)
            idx -= n1, l |= 1;
        if (idx >= n2)
            idx -= n2, l |= 2;
        if (idx >= n3)
            idx -= n3, l |= 4;
        if (idx >= n4)
            idx -= n4, l |= 8;
        lane16 = l, slot = idx;
    }
    // v[0 .. N - 1) and `last` (sum N - 1; with out_last == nullptr v holds all N and `last` is ignored): this thread's partial
    // sums; the totals go to out[0 .. N - 1) and *out_last, the counters' totals to *count_a_out / *count_b_out.
    // stage: kStageDoubles doubles of LDS, cstage: [wavefronts][2] doubles.  (The first step reads the caller's accumulators
    // and writes a list of half the length: no second copy of the N accumulators is alive - with 46 of them, a homography's,
    // a copy cost the sweep of k_lm<3> its registers: +33 % on its point loop, measured.)
    __device__ static void run(const double *v, double last, uint32_t cnt_a, uint32_t cnt_b, double *stage, double (*cstage)[2], double *out,
                               double *out_last, uint32_t *count_a_out, uint32_t *count_b_out) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double s[n1];
        {
            const bool upper = lane & 1;
#pragma unroll
            for (int i = 0; i < n1; ++i) {
                const double a = v[i];
                const double b = (i + n1 < N) ? ((i + n1 == N - 1 && out_last) ? last : v[i + n1]) : 0.0;
                const double keep = upper ? b : a, send = upper ? a : b;
                s[i] = keep + xchg<0>(send);
            }
        }
        halve<1, n1>(s, lane);
        halve<2, n2>(s, lane);
        halve<3, n3>(s, lane);
#pragma unroll
        for (int i = 0; i < n4; ++i)
            stage[(wave * n4 + i) * 64 + lane] = s[i];
        const uint32_t ca = wave_sum_u32(cnt_a), cb = wave_sum_u32(cnt_b);
        if (lane == 0) {
            cstage[wave][0] = (double)ca;
            cstage[wave][1] = (double)cb;
        }
        __syncthreads();
        if (threadIdx.x < N) {
            int l16, slot;
            home((int)threadIdx.x, l16, slot);
            double wv[kLMThreads / 64];
#pragma unroll
            for (int w = 0; w < kLMThreads / 64; ++w) { // (all loads first: the rows of the wavefronts are independent)
                const double *r = stage + (w * n4 + slot) * 64 + l16;
                wv[w] = (r[48] + r[32]) + (r[16] + r[0]); // rows 3 + 2, rows 1 + 0: the butterfly's row_bcast:15 / :31 steps
            }
            double t = 0;
#pragma unroll
            for (int w = 0; w < kLMThreads / 64; ++w)
                t += wv[w];
            if (threadIdx.x < N - 1 || !out_last)
                out[threadIdx.x] = t;
            else
                *out_last = t;
        } else if (threadIdx.x < N + 2) {
            const int k = (int)threadIdx.x - N;
            double t = 0;
            for (int w = 0; w < kLMThreads / 64; ++w)
                t += cstage[w][k];
            *(k == 0 ? count_a_out : count_b_out) = (uint32_t)t;
        }
        __syncthreads();
    }
};

// `lds_points` != 0: the correspondences are staged once into dynamic LDS (nd * n doubles) and every residual /
// Jacobian pass reads them from there - a task makes 30..50 passes over the same points, and for the small problems
// of the default-options regime the L2 round trip of every pass is a visible part of the 8 us an LM iteration takes.
#ifdef PL_LM_PROFILE // experiment builds (scripts/exp): cycles of thread 0 per phase of k_lm, summed over tasks; read by pl_debug_lm_profile
__device__ unsigned long long g_lm_prof[16];
#define PL_PROF_T0() const unsigned long long prof_t0_ = __builtin_readcyclecounter()
#define PL_PROF_ADD(slot, t0) do { if (threadIdx.x == 0) atomicAdd(&g_lm_prof[slot], __builtin_readcyclecounter() - (t0)); } while (0)
#else
#define PL_PROF_T0() do { } while (0)
#define PL_PROF_ADD(slot, t0) do { } while (0)
#endif
template <int EST> __global__ __launch_bounds__(kLMThreads) void k_lm(LMTask *tasks, uint32_t lds_bytes) {
    using R = Refiner<EST>;
    constexpr int K = R::K;
    constexpr int NT = NormalSize<K>::kTotal;
    constexpr int ND = point_doubles(EST);
    // The task may live in pinned host memory the device reads over the bus (no upload dispatch): ONE cooperative
    // fetch into LDS; the outputs go back to the task (and the refined model's record to device memory) at the end.
    __shared__ LMTask s_task;
    LMTask &Tout = tasks[blockIdx.x];
    {
        static_assert(sizeof(LMTask) % 8 == 0, "copied as 64-bit words");
        const uint64_t *src = reinterpret_cast<const uint64_t *>(&Tout);
        uint64_t *dst = reinterpret_cast<uint64_t *>(&s_task);
        for (uint32_t w = threadIdx.x; w < sizeof(LMTask) / 8; w += kLMThreads)
            dst[w] = src[w];
        __syncthreads();
    }
    const LMTask &T = s_task;
    auto finish = [&](const double *params, bool skipped, uint32_t iterations, double cost, double initial_cost) {
        // (thread 0 only)
        for (int i = 0; i < kParamDoubles; ++i)
            Tout.params[i] = params[i];
        Tout.iterations = iterations;
        Tout.skipped = skipped ? 1u : 0u;
        Tout.cost = cost;
        Tout.initial_cost = initial_cost;
        if (T.record_out) {
            if (skipped) { // refinement not run: model unchanged (relative_pose.cc:75-77)
                for (int i = 0; i < kModelStride; ++i)
                    T.record_out[i] = T.record_in[i];
            } else {
                record_from_lm_params(EST, params, T.record_out);
            }
        }
    };
    extern __shared__ double s_lm_points[];
    const PointSet pts_global = T.pts;
    PointSet pts = pts_global;
    // (tasks of several problems may share a launch: each stages its own points if they fit the launch's dynamic LDS)
    const bool lds_points = sizeof(double) * ND * (size_t)pts_global.n <= (size_t)lds_bytes;
    if (lds_points) {
        for (int d = 0; d < ND; ++d) {
            for (uint32_t i = threadIdx.x; i < pts_global.n; i += kLMThreads)
                s_lm_points[(size_t)d * pts_global.n + i] = pts_global.a[d][i];
            pts.a[d] = s_lm_points + (size_t)d * pts_global.n;
        }
    }

    __shared__ LMControl ctl;
    __shared__ double cur[kParamDoubles], trial[kParamDoubles];
    __shared__ RefineCtx ctx;
    __shared__ double scratch[kLMThreads / 64][NT + 1];
    __shared__ double tr_stage[BlockReduceT<NT + 1>::kStageDoubles];
    __shared__ double tr_counts[kLMThreads / 64][2];
    __shared__ double normal[NT];
    __shared__ double normal_next[NT]; // the normal equations at the trial point (fused pass), the next iteration's if the step is accepted
    __shared__ double s_racc[1];
    __shared__ uint32_t s_count;   // residual pass: correspondences counted (jacobian_accumulator.h's single counter after residual())
    __shared__ uint32_t s_count_j; // Jacobian pass: correspondences with a non-zero weight (... after accumulate())
    __shared__ int s_accepted;
    __shared__ int s_skip;
    __shared__ uint32_t s_queue[kLMThreads / 64][128]; // per wavefront: correspondences waiting for their Jacobian
    __shared__ __attribute__((aligned(16))) double s_terms[132][NT + 1]; // small problems: the terms of 64 correspondences (x 2 for H) + cost; two column-major buffers of 64 rows for the others

    const uint8_t *mask = T.mask;
    const double pscale = T.point_scale;
    const CameraParams cam = T.cam;

    if (T.gate_count && *T.gate_count <= T.gate_min) { // (uniform: every thread reads the same word) the task does not run
        if (threadIdx.x == 0) {
            Tout.iterations = 0;
            Tout.skipped = 2u;
        }
        return;
    }
    if (threadIdx.x == 0) {
        lm_start_params(EST, T, cur);
        ctl.opt = T.opt;
        ctl.loss = make_loss(T.opt.loss_type, T.opt.loss_scale);
        ctl.done = 0;
        s_skip = 0;
    }
    __syncthreads();

    // ---- relative-pose LO: restrict to approximate inliers (relative_pose.cc:62-86) ----
    if constexpr (EST == EST_REL) {
        if (T.prefilter_thr2 > 0) {
            double M[kModelStride];
            {
                Quat q;
                q.w = cur[0], q.x = cur[1], q.y = cur[2], q.z = cur[3];
                store_pose_model_q(M, q, v3(cur[4], cur[5], cur[6]), true);
            }
            uint32_t c = 0;
            for (uint32_t i = threadIdx.x; i < pts.n; i += kLMThreads) {
                double r2;
                const bool in = sampson_pose_inlier(M, pts.a[0][i], pts.a[1][i], pts.a[2][i], pts.a[3][i],
                                                    T.prefilter_thr2, r2);
                T.scratch[i] = in ? 1 : 0;
                c += in;
            }
            double dummy[1] = {0.0};
            __shared__ double pre_scratch[kLMThreads / 64][2];
            __shared__ double pre_out[1];
            BlockReduce<1>::run(dummy, c, pre_scratch, pre_out, &s_count);
            if (threadIdx.x == 0 && s_count <= 5)
                s_skip = 1;
            __threadfence_block();
            __syncthreads();
            mask = T.scratch;
        }
    }
    if (s_skip) {
        if (threadIdx.x == 0)
            finish(cur, true, 0u, 0.0, 0.0);
        return;
    }

    // One pass over the points.  mode kRes: robust cost only (-> s_racc, s_count).  kJac: normal equations (-> out, s_count_j).
    // kBoth: both at the same parameters, every sum in the order of the separate passes - the cost of a trial step and, if the
    // step is accepted, the next iteration's normal equations from ONE sweep (lm_impl.h:88-99 then :66-76 of the next
    // iteration evaluate the same point; k_lm2 does the same across launches).
    enum { kRes = 0, kJac = 1, kBoth = 2 };
    auto pass = [&](const double *p, int mode, double *out) {
        const bool jac = mode != kRes, res = mode != kJac;
#ifdef PL_LM_PROFILE
        const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
        if (threadIdx.x == 0) {
            R::prepare(p, ctx);
        }
        __syncthreads();
#ifdef PL_LM_PROFILE
        const unsigned long long pt1 = __builtin_readcyclecounter();
        PL_PROF_ADD(0, pt0); // prepare + barrier
#endif
        double acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
            acc[i] = 0.0;
        double racc = 0.0;
        uint32_t cnt = 0, cntj = 0; // residual pass's counter / Jacobian pass's counter
        const Loss loss = ctl.loss;
        // One correspondence into the normal equations (jac) or into the robust cost (!jac) of this thread's accumulators
        // (named directly - handed over as pointers they would live in scratch memory)
        auto point = [&](uint32_t i, bool jacobian_pass) {
            if constexpr (EST == EST_ABS) {
                const double x = pts.a[0][i] * pscale, y = pts.a[1][i] * pscale;
                const double X = pts.a[2][i], Y = pts.a[3][i], Z = pts.a[4][i];
                double r0, r1;
                if (!jacobian_pass) {
                    if (R::residual(p, ctx, cam, x, y, X, Y, Z, r0, r1)) {
                        racc += 1.0 * loss_value(loss, r0 * r0 + r1 * r1);
                        cnt++;
                    }
                } else {
                    double J[2 * K];
                    if (R::jacobian(p, ctx, cam, x, y, X, Y, Z, r0, r1, J))
                        accumulate2<K>(acc, loss, r0, r1, J, cntj);
                }
            } else if constexpr (EST == EST_HOM) {
                const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
                double f0, f1, g0, g1;
                if (!jacobian_pass) {
                    R::residual(ctx, a0, a1, b0, b1, f0, f1, g0, g1);
                    racc += 1.0 * loss_value(loss, f0 * f0 + f1 * f1);
                    racc += 1.0 * loss_value(loss, g0 * g0 + g1 * g1);
                    cnt += 2;
                } else {
                    double Jf[2 * K], Jb[2 * K];
                    R::jacobian(ctx, a0, a1, b0, b1, f0, f1, Jf, g0, g1, Jb);
                    accumulate2<K>(acc, loss, f0, f1, Jf, cntj);
                    accumulate2<K>(acc, loss, g0, g1, Jb, cntj);
                }
            } else {
                const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
                if (!jacobian_pass) {
                    const double r = R::residual(ctx, a0, a1, b0, b1);
                    racc += 1.0 * loss_value(loss, r * r);
                    cnt++;
                } else {
                    double J[K];
                    const double r = R::jacobian(ctx, a0, a1, b0, b1, J);
                    accumulate1<K>(acc, loss, r, J, cntj);
                }
            }
        };
        // Small problems (n <= kLMSeqPoints): the sums in the REFERENCE's order.  With a handful of correspondences the
        // models of a run tie exactly in their MSAC score (minimal support: every inlier is a sample point, the residuals
        // vanish against (N - count) thr^2), and which of two tied LO results wins is decided by the last bit of the refined
        // model - i.e. by the order the normal equations are summed in.  The reference adds correspondence after
        // correspondence (jacobian_accumulator.h:82-97); here every thread forms its correspondence's terms, and thread a
        // adds term a of all correspondences one after the other, 64 correspondences per round through LDS.  (x + 0.0 = x:
        // the terms of skipped correspondences are zeros.)
        if (pts.n <= (uint32_t)kLMSeqPoints) {
            constexpr int SUB = (EST == EST_HOM) ? 2 : 1;
            double term[SUB][NT], cterm[SUB];
#pragma unroll
            for (int u = 0; u < SUB; ++u) {
                cterm[u] = 0.0;
#pragma unroll
                for (int a = 0; a < NT; ++a)
                    term[u][a] = 0.0;
            }
            uint32_t cn = 0, cnj = 0;
            // (the same expressions as `point`, into this correspondence's own terms; the homography's backward block apart)
                auto point_terms = [&](uint32_t i, bool jacobian_pass) {
                if constexpr (EST == EST_ABS) {
                    const double x = pts.a[0][i] * pscale, y = pts.a[1][i] * pscale;
                    const double X = pts.a[2][i], Y = pts.a[3][i], Z = pts.a[4][i];
                    double r0, r1;
                    if (!jacobian_pass) {
                        if (R::residual(p, ctx, cam, x, y, X, Y, Z, r0, r1)) {
                            cterm[0] += 1.0 * loss_value(loss, r0 * r0 + r1 * r1);
                            cn++;
                        }
                    } else {
                        double J[2 * K];
                        if (R::jacobian(p, ctx, cam, x, y, X, Y, Z, r0, r1, J))
                            accumulate2<K>(term[0], loss, r0, r1, J, cnj);
                    }
                } else if constexpr (EST == EST_HOM) {
                    const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
                    double f0, f1, g0, g1;
                    if (!jacobian_pass) {
                        R::residual(ctx, a0, a1, b0, b1, f0, f1, g0, g1);
                        cterm[0] += 1.0 * loss_value(loss, f0 * f0 + f1 * f1);
                        cterm[SUB - 1] += 1.0 * loss_value(loss, g0 * g0 + g1 * g1);
                        cn += 2;
                    } else {
                        double Jf[2 * K], Jb[2 * K];
                        R::jacobian(ctx, a0, a1, b0, b1, f0, f1, Jf, g0, g1, Jb);
                        accumulate2<K>(term[0], loss, f0, f1, Jf, cnj);
                        accumulate2<K>(term[SUB - 1], loss, g0, g1, Jb, cnj);
                    }
                } else {
                    const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
                    if (!jacobian_pass) {
                        const double r = R::residual(ctx, a0, a1, b0, b1);
                        cterm[0] += 1.0 * loss_value(loss, r * r);
                        cn++;
                    } else {
                        double J[K];
                        const double r = R::jacobian(ctx, a0, a1, b0, b1, J);
                        accumulate1<K>(term[0], loss, r, J, cnj);
                    }
                }
            };
            if (threadIdx.x < pts.n && !(mask && !mask[threadIdx.x])) {
                if (res)
                    point_terms(threadIdx.x, false);
                if (jac)
                    point_terms(threadIdx.x, true);
            }
            if (threadIdx.x == 0) {
                if (res)
                    s_count = 0;
                if (jac)
                    s_count_j = 0;
            }
            __syncthreads();
            if (cn)
                atomicAdd(&s_count, cn);
            if (cnj)
                atomicAdd(&s_count_j, cnj);
            double tot = 0.0;
            const uint32_t rounds = (pts.n + 63u) / 64u;
            if constexpr (SUB == 1) {
                // Two buffers of 64 rows, column-major [entry][row] (stride 66: the adders' columns spread over the banks): wavefront rd
                // writes its correspondences' terms while the adder lanes of wavefront 0 add round rd - 1 with the inline-asm chain of
                // k_lm_ordered (pl_lm_chain.inc: 11.7 cycles per row; the rows beyond n are zeros: x + 0.0 = x) - one barrier per round.
                // Before (cycle counters, n = 200): write, barrier, 64 rows at ~20 cycles, barrier: 14.4 k cycles per pass, 58 % of an LM
                // iteration of the latency configuration (BASELINE configs[0]).
                constexpr int kSeqStride = 66;
                double *const tb = &s_terms[0][0];
                static_assert(2 * (NT + 1) * kSeqStride <= (int)(sizeof(s_terms) / sizeof(double)), "two buffers of 64 rows");
                const bool adder = (jac && threadIdx.x < NT) || (res && threadIdx.x == NT);
                for (uint32_t rd = 0; rd <= rounds; ++rd) {
                    if (rd < rounds && (threadIdx.x >> 6) == rd) {
                        double *dst = tb + (size_t)(rd & 1u) * (NT + 1) * kSeqStride + (threadIdx.x & 63);
                        if (jac) {
#pragma unroll
                            for (int a = 0; a < NT; ++a)
                                dst[a * kSeqStride] = term[0][a];
                        }
                        if (res)
                            dst[NT * kSeqStride] = cterm[0];
                    }
                    if (rd > 0 && adder) {
                        const uint32_t addr = (uint32_t)(uintptr_t)(tb + (size_t)((rd - 1u) & 1u) * (NT + 1) * kSeqStride + (size_t)threadIdx.x * kSeqStride);
                        PL_LM_CHAIN64(tot, addr);
                    }
                    __syncthreads();
                }
                if (jac && threadIdx.x < NT)
                    out[threadIdx.x] = tot;
                if (res && threadIdx.x == NT)
                    s_racc[0] = tot;
                __syncthreads();
                return;
            }
            for (uint32_t rd = 0; rd < rounds; ++rd) {
                if ((threadIdx.x >> 6) == rd) {
#pragma unroll
                    for (int u = 0; u < SUB; ++u) {
                        double *dst = s_terms[SUB * (threadIdx.x & 63) + u];
                        if (jac) {
#pragma unroll
                            for (int a = 0; a < NT; ++a)
                                dst[a] = term[u][a];
                        }
                        if (res)
                            dst[NT] = cterm[u];
                    }
                }
                __syncthreads();
                const uint32_t slots = min(64u, pts.n - 64u * rd) * SUB;
                if ((jac && threadIdx.x < NT) || (res && threadIdx.x == NT)) {
                    // (eight LDS reads travel together, the additions stay a chain in correspondence order)
                    uint32_t q = 0;
                    for (; q + 8u <= slots; q += 8u) {
                        double t8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            t8[u] = s_terms[q + u][threadIdx.x];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            tot += t8[u];
                    }
                    for (; q < slots; ++q)
                        tot += s_terms[q][threadIdx.x];
                }
                __syncthreads();
            }
            if (jac && threadIdx.x < NT)
                out[threadIdx.x] = tot;
            if (res && threadIdx.x == NT)
                s_racc[0] = tot;
            __syncthreads();
            return;
        }
        // Truncated losses (the LO's: bundle.cc TRUNCATED at max_error) give weight zero to every correspondence beyond the
        // threshold - 70 % of them in config 1 - but a Jacobian costs 10x a residual and a wavefront pays for it as soon as
        // ONE lane holds an inlier.  So the Jacobian pass first evaluates the residuals only, queues the correspondences
        // with a non-zero weight per wavefront (ballot + prefix, ascending order) and runs Jacobian + accumulation on full
        // wavefronts of them.  Which lane accumulates a correspondence changes, the set of terms does not; the reduction
        // over lanes and wavefronts below is in fixed order as before.
        const bool zero_weights = loss.type == LOSS_TRUNCATED || loss.type == LOSS_TRUNCATED_CAUCHY;
        if (jac && zero_weights && !(EST == EST_REL && mask)) {
            const int lane = threadIdx.x & 63;
            uint32_t *const q = s_queue[threadIdx.x >> 6];
            uint32_t qn = 0; // wave-uniform: correspondences waiting, q[0 .. qn)
            for (uint32_t base = (threadIdx.x >> 6) * 64u; base < pts.n; base += kLMThreads) {
                const uint32_t i = base + (uint32_t)lane;
                bool keep = false;
                if (i < pts.n && !(mask && !mask[i])) { // (res: the residual pass's terms, in its per-lane order)
                    if constexpr (EST == EST_ABS) {
                        double r0, r1;
                        const bool valid = R::residual(p, ctx, cam, pts.a[0][i] * pscale, pts.a[1][i] * pscale, pts.a[2][i],
                                                       pts.a[3][i], pts.a[4][i], r0, r1);
                        keep = valid && loss_weight(loss, r0 * r0 + r1 * r1) != 0;
                        if (res && valid) {
                            racc += 1.0 * loss_value(loss, r0 * r0 + r1 * r1);
                            cnt++;
                        }
                    } else if constexpr (EST == EST_HOM) {
                        double f0, f1, g0, g1;
                        R::residual(ctx, pts.a[0][i], pts.a[1][i], pts.a[2][i], pts.a[3][i], f0, f1, g0, g1);
                        keep = loss_weight(loss, f0 * f0 + f1 * f1) != 0 || loss_weight(loss, g0 * g0 + g1 * g1) != 0;
                        if (res) {
                            racc += 1.0 * loss_value(loss, f0 * f0 + f1 * f1);
                            racc += 1.0 * loss_value(loss, g0 * g0 + g1 * g1);
                            cnt += 2;
                        }
                    } else {
                        const double r = R::residual(ctx, pts.a[0][i], pts.a[1][i], pts.a[2][i], pts.a[3][i]);
                        keep = loss_weight(loss, r * r) != 0;
                        if (res) {
                            racc += 1.0 * loss_value(loss, r * r);
                            cnt++;
                        }
                    }
                }
                const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
                if (m) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (keep)
                        q[qn + below] = i;
                    qn += (uint32_t)__popcll(m);
                    if (qn >= 64u) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        point(q[lane], true);
                        const uint32_t rest = qn - 64u;
                        const uint32_t moved = ((uint32_t)lane < rest) ? q[64 + lane] : 0u;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if ((uint32_t)lane < rest)
                            q[lane] = moved;
                        qn = rest;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if ((uint32_t)lane < qn)
                point(q[lane], true);
        } else {
            for (uint32_t i = threadIdx.x; i < pts.n; i += kLMThreads) {
                if (mask && !mask[i])
                    continue;
                if (res)
                    point(i, false);
                if (jac)
                    point(i, true);
            }
        }
#ifdef PL_LM_PROFILE
        const unsigned long long pt2 = __builtin_readcyclecounter();
        PL_PROF_ADD(1, pt1); // the sweep over the points (thread 0's share)
        __syncthreads();
        const unsigned long long pt3 = __builtin_readcyclecounter();
        PL_PROF_ADD(2, pt2); // waiting for the slowest wavefront of the sweep
#endif
#ifdef PL_LM_OLD_REDUCE // experiment builds: round 4's reductions (A/B on one box)
        if (res) {
            double v[1] = {racc};
            BlockReduce<1>::run(v, cnt, reinterpret_cast<double(*)[2]>(&scratch[0][0]), s_racc, &s_count);
        }
        if (jac)
            BlockReduce<NT>::run(acc, cntj, scratch, out, &s_count_j);
#else
        if (res && jac) { // the fused sweep: the NT sums of the normal equations and the cost through one reduction
            BlockReduceT<NT + 1>::run(acc, racc, cntj, cnt, tr_stage, tr_counts, out, &s_racc[0], &s_count_j, &s_count);
        } else if (res) {
            double v[1] = {racc};
            BlockReduce<1>::run(v, cnt, reinterpret_cast<double(*)[2]>(&scratch[0][0]), s_racc, &s_count);
        } else if (jac) {
            uint32_t unused;
            BlockReduceT<NT>::run(acc, 0.0, cntj, 0u, tr_stage, tr_counts, out, nullptr, &s_count_j, &unused);
        }
#endif
#ifdef PL_LM_PROFILE
        PL_PROF_ADD(3, pt3); // block reductions
        if (threadIdx.x == 0)
            atomicAdd(&g_lm_prof[8], 1ull); // passes
#endif
    };

    // The initial cost and the first iteration's normal equations are evaluated at the same point with the same loss: one sweep.
    if constexpr (EST == EST_REL) {
        if (threadIdx.x == 0)
            R::prepare_params(cur);
        __syncthreads();
    }
    const bool first_needs_jacobian = T.opt.max_iterations != 0;
    pass(cur, first_needs_jacobian ? kBoth : kRes, normal);
    if (threadIdx.x == 0)
        lm_begin(ctl, T.opt, s_racc[0], s_count);
    __syncthreads();

    // The trial point's sweep is fused (kBoth) unless the loss changes between iterations (TRUNCATED_LE_ZACH: mu grows after
    // every iteration, bundle.cc:52-75 - the next Jacobian would have to be evaluated with the new mu).
    const bool fuse = T.opt.loss_type != LOSS_TRUNCATED_LE_ZACH;
    bool have_next = first_needs_jacobian; // `normal` already holds the normal equations at `cur`
    uint32_t jac_count = s_count_j;        // the Jacobian pass's counter that belongs to `normal`
    while (!ctl.done) {
        const bool fresh = ctl.rejac != 0;
        if (fresh && !have_next) {
            if constexpr (EST == EST_REL) {
                if (threadIdx.x == 0)
                    R::prepare_params(cur);
                __syncthreads();
            }
            pass(cur, kJac, normal);
            jac_count = s_count_j;
        }
        // (the one lane that solves is a chain of dependent fp64 operations: with other kernels' wavefronts on the SIMD it gets an issue slot
        // every few instructions only - priority 3 for the serial sections, measured on the mixed batch)
#ifdef PL_LM_PROFILE
        const unsigned long long st0 = __builtin_readcyclecounter();
#endif
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(3);
        if (threadIdx.x == 0) {
            lm_solve<K>(ctl, normal, fresh, jac_count);
#ifdef PL_LM_PROFILE
            atomicAdd(&g_lm_prof[4], __builtin_readcyclecounter() - st0); // lm_solve alone
#endif
            if (!ctl.done) {
                R::step(cur, ctx, ctl.sol, trial);
                if constexpr (EST == EST_REL)
                    if (fuse)
                        R::prepare_params(trial); // (the tangent basis of the Jacobian at the trial point: what the next
                                                  // iteration computes from the accepted parameters)
            }
        }
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(0);
        __syncthreads();
#ifdef PL_LM_PROFILE
        PL_PROF_ADD(5, st0); // solve + step (+ prepare_params) + barrier
#endif
        if (ctl.done)
            break;
        pass(trial, fuse ? kBoth : kRes, normal_next);
#ifdef PL_LM_PROFILE
        const unsigned long long ut0 = __builtin_readcyclecounter();
#endif
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(3);
        if (threadIdx.x == 0) {
            const bool accepted = lm_update<K>(ctl, normal, s_racc[0], s_count);
            if (accepted)
                for (int i = 0; i < kParamDoubles; ++i)
                    cur[i] = trial[i];
            s_accepted = accepted ? 1 : 0;
        }
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        have_next = fuse && s_accepted != 0;
        if (have_next) {
            if (threadIdx.x < NT)
                normal[threadIdx.x] = normal_next[threadIdx.x];
            jac_count = s_count_j;
            __syncthreads();
        }
#ifdef PL_LM_PROFILE
        PL_PROF_ADD(6, ut0); // update + copy + barriers
        if (threadIdx.x == 0)
            atomicAdd(&g_lm_prof[9], 1ull); // iterations
#endif
    }

    if (threadIdx.x == 0)
        finish(cur, false, ctl.iterations, ctl.cost, ctl.initial_cost);
}

// ---- k_lm_ordered: the same refinement with EVERY sum in the reference's order, for every n (round 4) -------------------
// k_lm above adds the normal equations and the robust cost of problems beyond 256 correspondences in tree order (per-lane partial
// sums, wave and block reduction): refined models 1e-13 off the reference's, which decides a comparison only when two values tie
// to that level - never observed, but not excluded.  This kernel excludes it: the opt-in exact mode (pl_set_lm_mode(1) /
// POSELIB_AMD_LM_ORDERED=1) routes every task through it.  Cost and floor (measured, DESIGN 4 "Ordered sums at every n"): a
// sequential fp64 sum is a chain of dependent additions, 9.3 cycles each on gfx950; the consumer below reaches 11.7 cycles per row in
// isolation and ~21 inside the kernel, i.e. 44 us per LM iteration at n = 5000 against 22 us for the tree form.
#ifndef PL_LM_WAVES
#define PL_LM_WAVES 2 // wavefronts per SIMD the register allocation of k_lm aims at (2: one workgroup per CU, no spills)
#endif
template <int EST> __global__ __launch_bounds__(kLMThreads, (EST == EST_HOM ? 2 : PL_LM_WAVES)) void k_lm_ordered(LMTask *tasks, uint32_t lds_bytes) {
    using R = Refiner<EST>;
    constexpr int K = R::K;
    constexpr int NT = NormalSize<K>::kTotal;
    constexpr int ND = point_doubles(EST);
    // The task may live in pinned host memory the device reads over the bus (no upload dispatch): ONE cooperative
    // fetch into LDS; the outputs go back to the task (and the refined model's record to device memory) at the end.
    __shared__ LMTask s_task;
    LMTask &Tout = tasks[blockIdx.x];
    {
        static_assert(sizeof(LMTask) % 8 == 0, "copied as 64-bit words");
        const uint64_t *src = reinterpret_cast<const uint64_t *>(&Tout);
        uint64_t *dst = reinterpret_cast<uint64_t *>(&s_task);
        for (uint32_t w = threadIdx.x; w < sizeof(LMTask) / 8; w += kLMThreads)
            dst[w] = src[w];
        __syncthreads();
    }
    const LMTask &T = s_task;
    auto finish = [&](const double *params, bool skipped, uint32_t iterations, double cost, double initial_cost) {
        // (thread 0 only)
        for (int i = 0; i < kParamDoubles; ++i)
            Tout.params[i] = params[i];
        Tout.iterations = iterations;
        Tout.skipped = skipped ? 1u : 0u;
        Tout.cost = cost;
        Tout.initial_cost = initial_cost;
        if (T.record_out) {
            if (skipped) { // refinement not run: model unchanged (relative_pose.cc:75-77)
                for (int i = 0; i < kModelStride; ++i)
                    T.record_out[i] = T.record_in[i];
            } else {
                record_from_lm_params(EST, params, T.record_out);
            }
        }
    };
    extern __shared__ double s_lm_points[];
    const PointSet pts_global = T.pts;
    PointSet pts = pts_global;
    // (tasks of several problems may share a launch: each stages its own points if they fit the launch's dynamic LDS)
    const bool lds_points = sizeof(double) * ND * (size_t)pts_global.n <= (size_t)lds_bytes;
    if (lds_points) {
        for (int d = 0; d < ND; ++d) {
            for (uint32_t i = threadIdx.x; i < pts_global.n; i += kLMThreads)
                s_lm_points[(size_t)d * pts_global.n + i] = pts_global.a[d][i];
            pts.a[d] = s_lm_points + (size_t)d * pts_global.n;
        }
    }

    __shared__ LMControl ctl;
    __shared__ double cur[kParamDoubles], trial[kParamDoubles];
    __shared__ RefineCtx ctx;
    __shared__ double normal[NT];
    __shared__ double normal_next[NT]; // the normal equations at the trial point (fused pass), the next iteration's if the step is accepted
    __shared__ double s_racc[1];
    __shared__ uint32_t s_count;   // residual pass: correspondences counted (jacobian_accumulator.h's single counter after residual())
    __shared__ uint32_t s_count_j; // Jacobian pass: correspondences with a non-zero weight (... after accumulate())
    __shared__ int s_accepted;
    __shared__ int s_skip;
    // Ring of term rows between the producer wavefronts (1 .. 7) and the consumer (wavefront 0), see `pass`: one slot = the rows of
    // 64 correspondences (x 2 for the homography's forward / backward blocks), a row = the NT entries of [J^T J lower triangle | J^T r]
    // followed by the correspondence's robust-cost term.  Measured and dropped (n = 5000, us per LM iteration; this form: 44): slots
    // of 128 rows filled in two producer rounds (55.6: a producer holds its slot twice as long, the consumer waits), the next
    // slot's flag read inside the consumer's asm block (47.5: the flag is rarely ahead), 15 producers (46.6), two workgroups per
    // CU at 128 registers (44.2).  With producers that only write zero rows the pass still takes 37 us: the consumer's 17.8
    // cycles per row inside the workgroup (11.7 alone on a CU) are the bound, the producers' arithmetic adds 7.
    constexpr int SUB = (EST == EST_HOM) ? 2 : 1;
    constexpr int kRingSlots = (EST == EST_HOM) ? 2 : 3;
    // column-major: s_ring[slot][entry][row] - the consumer lane of an entry reads ITS rows as adjacent doubles (two per ds_read_b128);
    // the column stride 64 SUB + 2 doubles keeps the producers' row writes and the consumer's reads spread over the banks
    constexpr int kRows = 64 * SUB;
    constexpr int kColStride = kRows + 2;
    __shared__ __attribute__((aligned(16))) double s_ring[kRingSlots][NT + 1][kColStride];
    __shared__ uint32_t s_ready[kRingSlots]; // sequence number of the batch a slot holds
    __shared__ uint32_t s_consumed;          // sequence number of the last batch the consumer has added
    if (threadIdx.x < kRingSlots)
        s_ready[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        s_consumed = 0;
    uint32_t seq_base = 0; // batches handed through the ring so far (the same in every thread)

    const uint8_t *mask = T.mask;
    const double pscale = T.point_scale;
    const CameraParams cam = T.cam;

    if (T.gate_count && *T.gate_count <= T.gate_min) { // (uniform: every thread reads the same word) the task does not run
        if (threadIdx.x == 0) {
            Tout.iterations = 0;
            Tout.skipped = 2u;
        }
        return;
    }
    if (threadIdx.x == 0) {
        lm_start_params(EST, T, cur);
        ctl.opt = T.opt;
        ctl.loss = make_loss(T.opt.loss_type, T.opt.loss_scale);
        ctl.done = 0;
        s_skip = 0;
    }
    __syncthreads();

    // ---- relative-pose LO: restrict to approximate inliers (relative_pose.cc:62-86) ----
    if constexpr (EST == EST_REL) {
        if (T.prefilter_thr2 > 0) {
            double M[kModelStride];
            {
                Quat q;
                q.w = cur[0], q.x = cur[1], q.y = cur[2], q.z = cur[3];
                store_pose_model_q(M, q, v3(cur[4], cur[5], cur[6]), true);
            }
            uint32_t c = 0;
            for (uint32_t i = threadIdx.x; i < pts.n; i += kLMThreads) {
                double r2;
                const bool in = sampson_pose_inlier(M, pts.a[0][i], pts.a[1][i], pts.a[2][i], pts.a[3][i],
                                                    T.prefilter_thr2, r2);
                T.scratch[i] = in ? 1 : 0;
                c += in;
            }
            double dummy[1] = {0.0};
            __shared__ double pre_scratch[kLMThreads / 64][2];
            __shared__ double pre_out[1];
            BlockReduce<1>::run(dummy, c, pre_scratch, pre_out, &s_count);
            if (threadIdx.x == 0 && s_count <= 5)
                s_skip = 1;
            __threadfence_block();
            __syncthreads();
            mask = T.scratch;
        }
    }
    if (s_skip) {
        if (threadIdx.x == 0)
            finish(cur, true, 0u, 0.0, 0.0);
        return;
    }

    // One pass over the points.  mode kRes: robust cost only (-> s_racc, s_count).  kJac: normal equations (-> out, s_count_j).
    // kBoth: both at the same parameters - the cost of a trial step and, if the step is accepted, the next iteration's normal
    // equations from ONE sweep (lm_impl.h:88-99 then :66-76 of the next iteration evaluate the same point; k_lm2 does the same
    // across launches).
    //
    // EVERY sum is formed in the reference's order, for every n (round 4; up to round 3 only for n <= 256).  The reference adds
    // correspondence after correspondence (jacobian_accumulator.h:82-97: one `+=` per entry and correspondence), and which of two
    // tied or nearly tied refinements wins is decided by the last bits of those sums.  A sequential sum of n terms is a chain of
    // n dependent additions whatever the hardware, so the pass is a PIPELINE: wavefronts 1 .. 7 (producers) take the batches of
    // 64 consecutive correspondences round robin, evaluate residual / Jacobian and the entry TERMS of their correspondence in
    // registers, and hand them over as rows of an LDS ring; wavefront 0 (consumer) owns one entry per lane - lane a < NT entry a
    // of [J^T J | J^T r], lane NT the robust cost - and adds the rows of batch 0, 1, 2, ... one after the other (eight LDS reads
    // travel together, the additions stay a chain).  The producers' arithmetic hides behind the chain: ~10 cycles per
    // correspondence, 21 us per sweep at n = 5000.  Terms of skipped correspondences (masked out, behind the camera, weight
    // zero) are zeros: x + 0.0 = x.  Flags: s_ready[slot] = sequence number of the batch the slot holds (release / acquire at
    // workgroup scope), s_consumed = last batch added; a producer writes a slot once the batch `slots` earlier is consumed.
    enum { kRes = 0, kJac = 1, kBoth = 2 };
    constexpr uint32_t kProducers = kLMThreads / 64 - 1;
    auto pass = [&](const double *p, int mode, double *out) {
        const bool jac = mode != kRes, res = mode != kJac;
        if (threadIdx.x == 0) {
            R::prepare(p, ctx);
            if (res)
                s_count = 0;
            if (jac)
                s_count_j = 0;
        }
        __syncthreads();
        const Loss loss = ctl.loss;
        const int lane = threadIdx.x & 63;
        const uint32_t wave = threadIdx.x >> 6;
        const uint32_t n = pts.n;
        const uint32_t nbatch = (n + 63u) / 64u;
        if (wave == 0) {
            // ---- consumer ----
            const bool mine = (jac && lane < NT) || (res && lane == NT);
            const int col = mine ? lane : 0;
            double tot = 0.0;
            __builtin_amdgcn_s_setprio(3); // (the chain of dependent additions is the critical path of the pass)
            for (uint32_t j = 0; j < nbatch; ++j) {
                const uint32_t g = seq_base + j + 1u;
                const uint32_t slot = g % (uint32_t)kRingSlots;
                while (__hip_atomic_load(&s_ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != g)
                    __builtin_amdgcn_s_sleep(1);
                // every slot holds 64 * SUB rows (the lanes beyond n write zero rows: x + 0.0 = x), so the loop has a fixed trip
                // count: one inline-asm statement per slot (pl_lm_chain.inc, generated by scripts/gen_lm_chain.py) - the row reads
                // run six pairs ahead of the additions, 11.7 cycles per row against 9.3 for the bare chain of dependent additions
                // and 17 - 25 for hipcc's schedule of the same loop (scripts/exp/chain_add.cc)
                const uint32_t col_addr = (uint32_t)(uintptr_t)&s_ring[slot][col][0];
                if constexpr (SUB == 1)
                    PL_LM_CHAIN64(tot, col_addr);
                else
                    PL_LM_CHAIN128(tot, col_addr);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0)
                    __hip_atomic_store(&s_consumed, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_s_setprio(0);
            if (jac && lane < NT)
                out[lane] = tot;
            if (res && lane == NT)
                s_racc[0] = tot;
        } else {
            // ---- producers ----
            uint32_t cn = 0, cnj = 0;
            for (uint32_t j = wave - 1u; j < nbatch; j += kProducers) {
                const uint32_t g = seq_base + j + 1u;
                const uint32_t slot = g % (uint32_t)kRingSlots;
                const uint32_t i = 64u * j + (uint32_t)lane;
                const bool live = i < n && !(mask && !mask[i]);
                double cterm[SUB];
                double r0 = 0, r1 = 0, g0 = 0, g1 = 0;
                double J[(EST == EST_ABS || EST == EST_HOM) ? 2 * K : K], Jb[(EST == EST_HOM) ? 2 * K : 1];
                bool have = false; // this correspondence contributes a Jacobian row
#pragma unroll
                for (int u = 0; u < SUB; ++u)
                    cterm[u] = 0.0;
                if (live) {
                    if constexpr (EST == EST_ABS) {
                        const double x = pts.a[0][i] * pscale, y = pts.a[1][i] * pscale;
                        const double X = pts.a[2][i], Y = pts.a[3][i], Z = pts.a[4][i];
                        if (res) {
                            if (R::residual(p, ctx, cam, x, y, X, Y, Z, r0, r1)) {
                                cterm[0] += 1.0 * loss_value(loss, r0 * r0 + r1 * r1);
                                cn++;
                            }
                        }
                        if (jac)
                            have = R::jacobian(p, ctx, cam, x, y, X, Y, Z, r0, r1, J);
                    } else if constexpr (EST == EST_HOM) {
                        const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
                        if (res) {
                            R::residual(ctx, a0, a1, b0, b1, r0, r1, g0, g1);
                            cterm[0] += 1.0 * loss_value(loss, r0 * r0 + r1 * r1);
                            cterm[SUB - 1] += 1.0 * loss_value(loss, g0 * g0 + g1 * g1);
                            cn += 2;
                        }
                        if (jac) {
                            R::jacobian(ctx, a0, a1, b0, b1, r0, r1, J, g0, g1, Jb);
                            have = true;
                        }
                    } else {
                        const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
                        if (res) {
                            const double r = R::residual(ctx, a0, a1, b0, b1);
                            cterm[0] += 1.0 * loss_value(loss, r * r);
                            cn++;
                        }
                        if (jac) {
                            r0 = R::jacobian(ctx, a0, a1, b0, b1, J);
                            have = true;
                        }
                    }
                }
                // the slot is free once the batch `kRingSlots` earlier has been added
                if (g > (uint32_t)kRingSlots) {
                    while (__hip_atomic_load(&s_consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) + (uint32_t)kRingSlots < g)
                        __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int u = 0; u < SUB; ++u) {
                    const int row = SUB * lane + u;
                    if (jac) {
                        auto store = [&](int a, double v) { s_ring[slot][a][row] = v; };
                        if (have) {
                            if constexpr (EST == EST_ABS)
                                terms2<K>(loss, r0, r1, J, cnj, store);
                            else if constexpr (EST == EST_HOM) {
                                if (u == 0)
                                    terms2<K>(loss, r0, r1, J, cnj, store);
                                else
                                    terms2<K>(loss, g0, g1, Jb, cnj, store);
                            } else
                                terms1<K>(loss, r0, J, cnj, store);
                        } else {
#pragma unroll
                            for (int a = 0; a < NT; ++a)
                                store(a, 0.0);
                        }
                    }
                    if (res)
                        s_ring[slot][NT][row] = cterm[u];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0)
                    __hip_atomic_store(&s_ready[slot], g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (cn)
                atomicAdd(&s_count, cn);
            if (cnj)
                atomicAdd(&s_count_j, cnj);
        }
        seq_base += nbatch;
        __syncthreads();
    };

    // The initial cost and the first iteration's normal equations are evaluated at the same point with the same loss: one sweep.
    if constexpr (EST == EST_REL) {
        if (threadIdx.x == 0)
            R::prepare_params(cur);
        __syncthreads();
    }
    const bool first_needs_jacobian = T.opt.max_iterations != 0;
    pass(cur, first_needs_jacobian ? kBoth : kRes, normal);
    if (threadIdx.x == 0)
        lm_begin(ctl, T.opt, s_racc[0], s_count);
    __syncthreads();

    // The trial point's sweep is fused (kBoth) unless the loss changes between iterations (TRUNCATED_LE_ZACH: mu grows after
    // every iteration, bundle.cc:52-75 - the next Jacobian would have to be evaluated with the new mu).
    const bool fuse = T.opt.loss_type != LOSS_TRUNCATED_LE_ZACH;
    bool have_next = first_needs_jacobian; // `normal` already holds the normal equations at `cur`
    uint32_t jac_count = s_count_j;        // the Jacobian pass's counter that belongs to `normal`
    while (!ctl.done) {
        const bool fresh = ctl.rejac != 0;
        if (fresh && !have_next) {
            if constexpr (EST == EST_REL) {
                if (threadIdx.x == 0)
                    R::prepare_params(cur);
                __syncthreads();
            }
            pass(cur, kJac, normal);
            jac_count = s_count_j;
        }
        // (the one lane that solves is a chain of dependent fp64 operations: with other kernels' wavefronts on the SIMD it gets an issue slot
        // every few instructions only - priority 3 for the serial sections, measured on the mixed batch)
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(3);
        if (threadIdx.x == 0) {
            lm_solve<K>(ctl, normal, fresh, jac_count);
            if (!ctl.done) {
                R::step(cur, ctx, ctl.sol, trial);
                if constexpr (EST == EST_REL)
                    if (fuse)
                        R::prepare_params(trial); // (the tangent basis of the Jacobian at the trial point: what the next
                                                  // iteration computes from the accepted parameters)
            }
        }
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (ctl.done)
            break;
        pass(trial, fuse ? kBoth : kRes, normal_next);
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(3);
        if (threadIdx.x == 0) {
            const bool accepted = lm_update<K>(ctl, normal, s_racc[0], s_count);
            if (accepted)
                for (int i = 0; i < kParamDoubles; ++i)
                    cur[i] = trial[i];
            s_accepted = accepted ? 1 : 0;
        }
        if (threadIdx.x < 64)
            __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        have_next = fuse && s_accepted != 0;
        if (have_next) {
            if (threadIdx.x < NT)
                normal[threadIdx.x] = normal_next[threadIdx.x];
            jac_count = s_count_j;
            __syncthreads();
        }
    }

    if (threadIdx.x == 0)
        finish(cur, false, ctl.iterations, ctl.cost, ctl.initial_cost);
}

// ---- LM across several workgroups ---------------------------------------------------------------------------------
// k_lm keeps one refinement task on one CU; at N = 10^4 with 7 or 8 parameters that CU is VALU-saturated for ~70 us per
// LM iteration.  k_lm2 spreads the point range of every task over gridDim.x workgroups and turns the LM loop inside
// out: one launch = "advance the LM state with the partial sums the previous launch left, then run the next pass over
// my slice".  A pass evaluates the robust cost AND the normal equations at the same parameters (the trial point): if
// the step is accepted these are exactly the normal equations lm_impl.h would compute next, if it is rejected they are
// dropped and the previous ones are re-solved with the larger damping - so ONE launch per LM iteration.  Every
// workgroup of a task reduces the same partials in the same order and runs the same scalar state machine
// (pl_refine.h lm_begin / lm_solve / lm_update), so all of them arrive at the same state without talking to each
// other; slice 0 stores it.  States and partials are double-buffered by launch parity.  The host enqueues
// max_iterations + 2 launches; workgroups of finished tasks return at once.  No spinning, no inter-workgroup
// synchronisation inside a launch.
constexpr int kLM2Threads = 256;

struct LM2State {
    LMControl ctl;
    double cur[kParamDoubles], trial[kParamDoubles];
    double normal[44]; // reduced [tri | Jtr] at `cur`
    uint32_t count;    // residuals with non-zero weight behind `normal`
    int32_t phase;     // partials waiting for the next launch: 0 none, 1 pass at the start point, 3 pass at `trial`
    int32_t finished;
    int32_t pad;
};

template <int EST>
__global__ __launch_bounds__(kLM2Threads) void k_lm2(PointSet pts, LMTask *tasks, LM2State *states, double *partials,
                                                     int parity, int first) {
    using R = Refiner<EST>;
    constexpr int K = R::K;
    constexpr int NT = NormalSize<K>::kTotal;
    constexpr int NV = NT + 3; // + robust cost, + residual count, + count of non-zero weights
    constexpr int kWaves = kLM2Threads / 64;
    const uint32_t S = gridDim.x, T = gridDim.y, s = blockIdx.x, t = blockIdx.y;
    LMTask &task = tasks[t];
    LM2State *in = &states[(size_t)parity * T + t];
    LM2State *out = &states[(size_t)(1 - parity) * T + t];
    const double *pin = partials + ((size_t)parity * T + t) * S * NV;
    double *pout = partials + (((size_t)(1 - parity) * T + t) * S + s) * NV;

    __shared__ LM2State st;
    __shared__ RefineCtx ctx;
    __shared__ double sums[NV];
    __shared__ double scratch[kWaves][NV];
    __shared__ int s_pass; // 0 nothing, 1 pass at cur, 3 pass at trial

    if (threadIdx.x == 0) {
        if (first) {
            for (int i = 0; i < kParamDoubles; ++i)
                st.cur[i] = task.params[i], st.trial[i] = 0.0;
            st.ctl.opt = task.opt;
            st.ctl.loss = make_loss(task.opt.loss_type, task.opt.loss_scale);
            st.ctl.done = 0;
            st.count = 0;
            st.phase = 0;
            st.finished = 0;
        } else {
            st = *in;
        }
    }
    __syncthreads();
    if (st.finished)
        return;
    if (st.phase != 0 && threadIdx.x < NV) { // every workgroup of the task: same partials, same order
        double v = 0.0;
        for (uint32_t j = 0; j < S; ++j)
            v += pin[(size_t)j * NV + threadIdx.x];
        sums[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int next = 0;
        bool finish = false;
        const uint32_t cnt_res = (st.phase != 0) ? (uint32_t)sums[NT + 1] : 0u;
        const uint32_t cnt_jac = (st.phase != 0) ? (uint32_t)sums[NT + 2] : 0u;
        auto take_normal = [&]() {
            for (int i = 0; i < NT; ++i)
                st.normal[i] = sums[i];
            st.count = cnt_jac;
        };
        auto solve_and_step = [&](bool fresh) {
            lm_solve<K>(st.ctl, st.normal, fresh, st.count);
            if (st.ctl.done) {
                finish = true;
            } else {
                R::step(st.cur, ctx, st.ctl.sol, st.trial);
                next = 3;
                st.phase = 3;
            }
        };
        switch (st.phase) {
        case 0:
            next = 1;
            st.phase = 1;
            break;
        case 1: // cost and normal equations at the start point
            lm_begin(st.ctl, task.opt, sums[NT], cnt_res);
            if (st.ctl.done) {
                finish = true;
            } else {
                take_normal();
                solve_and_step(true);
            }
            break;
        default: { // cost and (speculative) normal equations at the trial point
            const bool accepted = lm_update<K>(st.ctl, st.normal, sums[NT], cnt_res); // gradient of the OLD point
            if (accepted) {
                for (int i = 0; i < kParamDoubles; ++i)
                    st.cur[i] = st.trial[i];
                take_normal();
            }
            if (st.ctl.done)
                finish = true;
            else
                solve_and_step(accepted);
            break;
        }
        }
        if (finish) {
            st.finished = 1;
            next = 0;
            if (s == 0) {
                for (int i = 0; i < kParamDoubles; ++i)
                    task.params[i] = st.cur[i];
                task.iterations = st.ctl.iterations;
                task.skipped = 0;
                task.cost = st.ctl.cost;
                task.initial_cost = st.ctl.initial_cost;
                in->finished = 1; // the launch after next reads this buffer again
            }
        }
        if (s == 0)
            *out = st;
        s_pass = next;
        if (next)
            R::prepare(next == 3 ? st.trial : st.cur, ctx);
    }
    __syncthreads();
    const int pass = s_pass;
    if (pass == 0)
        return;
    const double *p = (pass == 3) ? st.trial : st.cur;
    const uint8_t *mask = task.mask;
    const double pscale = task.point_scale;
    const CameraParams cam = task.cam;
    const Loss loss = st.ctl.loss;
    double acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
        acc[i] = 0.0;
    double racc = 0.0;
    uint32_t cnt = 0, cntj = 0;
    const uint32_t per = (pts.n + S - 1) / S;
    const uint32_t i0 = s * per, i1 = min(pts.n, i0 + per);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += kLM2Threads) {
        if (mask && !mask[i])
            continue;
        if constexpr (EST == EST_ABS) {
            const double x = pts.a[0][i] * pscale, y = pts.a[1][i] * pscale;
            const double X = pts.a[2][i], Y = pts.a[3][i], Z = pts.a[4][i];
            double r0, r1, J[2 * K];
            if (R::jacobian(p, ctx, cam, x, y, X, Y, Z, r0, r1, J)) {
                racc += 1.0 * loss_value(loss, r0 * r0 + r1 * r1);
                cnt++;
                accumulate2<K>(acc, loss, r0, r1, J, cntj);
            }
        } else if constexpr (EST == EST_HOM) {
            const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
            double f0, f1, g0, g1, Jf[2 * K], Jb[2 * K];
            R::jacobian(ctx, a0, a1, b0, b1, f0, f1, Jf, g0, g1, Jb);
            racc += 1.0 * loss_value(loss, f0 * f0 + f1 * f1);
            racc += 1.0 * loss_value(loss, g0 * g0 + g1 * g1);
            cnt += 2;
            accumulate2<K>(acc, loss, f0, f1, Jf, cntj);
            accumulate2<K>(acc, loss, g0, g1, Jb, cntj);
        } else {
            const double a0 = pts.a[0][i], a1 = pts.a[1][i], b0 = pts.a[2][i], b1 = pts.a[3][i];
            double J[K];
            const double r = R::jacobian(ctx, a0, a1, b0, b1, J);
            racc += 1.0 * loss_value(loss, r * r);
            cnt++;
            accumulate1<K>(acc, loss, r, J, cntj);
        }
    }
    // workgroup reduction in fixed order (DPP inside the wave, waves 0..3 in sequence), then this slice's partial
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const double v = wave_sum_dpp(acc[i]);
        if (lane == 0)
            scratch[wave][i] = v;
    }
    {
        const double v = wave_sum_dpp(racc);
        const uint32_t c = wave_sum_u32(cnt);
        const uint32_t cj = wave_sum_u32(cntj);
        if (lane == 0) {
            scratch[wave][NT] = v;
            scratch[wave][NT + 1] = (double)c;
            scratch[wave][NT + 2] = (double)cj;
        }
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double v = 0.0;
        for (int w = 0; w < kWaves; ++w)
            v += scratch[w][threadIdx.x];
        pout[threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------ launchers
#define PL_DISPATCH_EST(est, ...)                                                                                     \
    switch (est) {                                                                                                     \
    case EST_ABS: {                                                                                                    \
        constexpr int E = EST_ABS;                                                                                     \
        __VA_ARGS__;                                                                                                    \
    } break;                                                                                                           \
    case EST_REL: {                                                                                                    \
        constexpr int E = EST_REL;                                                                                     \
        __VA_ARGS__;                                                                                                    \
    } break;                                                                                                           \
    case EST_FUND: {                                                                                                   \
        constexpr int E = EST_FUND;                                                                                    \
        __VA_ARGS__;                                                                                                    \
    } break;                                                                                                           \
    case EST_HOM: {                                                                                                    \
        constexpr int E = EST_HOM;                                                                                     \
        __VA_ARGS__;                                                                                                    \
    } break;                                                                                                           \
    default:                                                                                                           \
        return hipErrorInvalidValue;                                                                                   \
    }

size_t generate_stage_bytes(int est, uint32_t num_iters) { return est == EST_REL ? rel_stage_bytes(num_iters) : 0; }
hipError_t launch_generate(int est, const GenerateArgs &a, hipStream_t stream) {
    if (a.num_iters == 0)
        return hipSuccess;
    const dim3 grid((a.num_iters + 63) / 64), block(64);
    if (est == EST_REL && a.stage) // four stages over a structure-of-arrays workspace (gen_rel.hip)
        return launch_generate_rel(a, stream);
    PL_DISPATCH_EST(est, k_generate<E><<<grid, block, 0, stream>>>(a));
    return hipGetLastError();
}

hipError_t launch_solve_batch(int est, const double *in, uint32_t np, double *models, uint32_t *num_models,
                              hipStream_t stream) {
    if (np == 0)
        return hipSuccess;
    const dim3 grid((np + 63) / 64), block(64);
    if (est == 4) {
        k_solve_essential<<<grid, block, 0, stream>>>(in, np, models, num_models);
        return hipGetLastError();
    }
    PL_DISPATCH_EST(est, k_solve_batch<E><<<grid, block, 0, stream>>>(in, np, models, num_models));
    return hipGetLastError();
}

// points per lane of the scorers (POSELIB_AMD_PF_P overrides, 1..6)
static int max_points_per_lane_pf() {
    static const int v = [] {
        const char *e = std::getenv("POSELIB_AMD_PF_P");
        const int x = e ? std::atoi(e) : 5;
        return x < 1 ? 1 : (x > 6 ? 6 : x);
    }();
    return v;
}
// `streaming`: launches of the batched main loop (k_score_queue: 64 * P correspondences per chunk, every wave of a
// workgroup sees the same chunk); otherwise the non-streaming kernels (256 * P correspondences per chunk).
// For the Sampson scores P is chosen to minimise  chunks(P) * (VALU issue slots of the pre-filter for P points + a
// per-hypothesis overhead): pairs of points share packed instructions, an odd point costs as much as a pair.
static void score_shape(int est, uint32_t n, bool streaming, bool mfma, uint32_t &chunks, int &P) {
    const uint32_t lanes = streaming ? 64u : (uint32_t)kScoreThreads;
    const int pmax = max_points_per_lane_pf();
    if (streaming && mfma && (est == EST_REL || est == EST_FUND)) {
        // k_score_mfma2: PG = 2 P groups of 32 correspondences per chunk; relative pose keeps the bearings in LDS as
        // well and stops at 320 correspondences per workgroup (two workgroups per CU)
        static const uint32_t m2p = [] { // POSELIB_AMD_M2_P (experiment): point groups of 64 per chunk of the Sampson matrix-core scorer
            const char *e = std::getenv("POSELIB_AMD_M2_P");
            return e ? (uint32_t)std::max(1, std::min(6, std::atoi(e))) : 0u;
        }();
        const uint32_t per_chunk_max = lanes * (m2p ? std::min<uint32_t>(m2p, est == EST_REL ? 5u : 6u) : (est == EST_REL ? 5u : 6u));
        chunks = std::max<uint32_t>(1u, (n + per_chunk_max - 1) / per_chunk_max);
        P = std::max<int>(1, (int)((n + lanes * chunks - 1) / (lanes * chunks)));
        return;
    }
    if (streaming && (est == EST_REL || est == EST_FUND) && std::getenv("POSELIB_AMD_PF_P") == nullptr) {
        // Sampson filter: 27 issue slots per pair of points, 29 for an odd one, ~30 per hypothesis around them
        // (measured on MI355X: P = 6 beats P = 5 by 25 % at N = 10000; for the cheaper reprojection and
        // homography filters the smaller register footprint of P = 5 wins)
        uint64_t best = ~0ull;
        for (int q = 1; q <= 6; ++q) {
            const uint64_t c = (n + lanes * q - 1) / (lanes * q);
            const uint64_t cost = (c ? c : 1) * (uint64_t)((q / 2) * 27 + (q & 1) * 29 + 30);
            if (cost < best)
                best = cost, P = q;
        }
        chunks = (n + lanes * P - 1) / (lanes * P);
        if (chunks == 0)
            chunks = 1;
        return;
    }
    // (absolute pose on the matrix-core form: at most 10 groups of 32 correspondences per chunk, see launch_score_est)
    const uint32_t per_chunk_max = lanes * (uint32_t)((streaming && mfma && (est == EST_ABS || est == EST_HOM)) ? std::min(pmax, 5) : pmax);
    chunks = (n + per_chunk_max - 1) / per_chunk_max;
    if (chunks == 0)
        chunks = 1;
    P = (int)((n + lanes * chunks - 1) / (lanes * chunks));
    if (P < 1)
        P = 1;
}
bool score_uses_mfma(int est, uint32_t n_points, const PrefilterArgs &pf) {
    static const bool off = std::getenv("POSELIB_AMD_NO_MFMA") != nullptr;
    static const bool off2 = std::getenv("POSELIB_AMD_NO_MFMA2") != nullptr;
    if (off || !pf.enabled || n_points < 1024u)
        return false;
    if (est == EST_ABS)
        return pf.g16 > 0.f && pf.thr <= 1.0f;
    if (est == EST_REL || est == EST_FUND)
        return !off2 && pf.t16 > 0.f; // (coordinates bounded by 8, threshold in range: make_prefilter_args)
    static const bool offh = std::getenv("POSELIB_AMD_NO_MFMAH") != nullptr;
    return !offh && pf.h16 > 0.f; // homography: the same conditions
}
uint32_t score_chunks(int est, uint32_t n, bool streaming, bool mfma) {
    uint32_t c;
    int P;
    score_shape(est, n, streaming, mfma, c, P);
    return c;
}

// slice count of a matrix-core scorer launch: a multiple of 8 from 8 on (slice_chunk_of_workgroup keeps a slice on one XCD)
static uint32_t xcd_slices(uint32_t s) { return std::max<uint32_t>(1u, s); }

template <int E>
static hipError_t launch_score_est(const ScoreArgs &a, uint32_t slices, hipStream_t stream) {
    uint32_t chunks;
    int P;
    const PrefilterArgs pf = a.pf;
    const bool streaming = (a.shadow && a.compact64) || a.shadow16; // batched main loop: hypothesis stream
    score_shape(E, a.pts.n, streaming, a.shadow16 != nullptr, chunks, P);
    if constexpr (E == EST_REL || E == EST_FUND) {
        if (streaming && a.shadow16) { // Sampson pre-filter on the matrix cores
            // two workgroups fit a CU (LDS): one resident round of 512 workgroups - a second, half-empty round would cost
            // as much as the first, and every workgroup pays the fp16 split of its chunk before its first hypothesis
            const uint32_t mslices = std::min<uint32_t>(slices * (uint32_t)kScoreThreads / (uint32_t)kMfmaThreads,
                                                        std::max<uint32_t>(1u, 512u / chunks));
            const dim3 mgrid(xcd_slices(mslices), chunks);
            const dim3 mblock(kMfmaThreads);
#define PL_M2_CASE(PP)                                                                                                 \
    case PP:                                                                                                           \
        k_score_mfma2<E, 2 * PP><<<mgrid, mblock, 0, stream>>>(a.pts, static_cast<const uint4 *>(a.shadow16), a.models, \
                                                              a.slots, a.num_hyp, a.hyp_capacity, a.thr2, pf,          \
                                                              a.part_count, a.part_score);                             \
        break;
            switch (P) {
                PL_M2_CASE(1)
                PL_M2_CASE(2)
                PL_M2_CASE(3)
                PL_M2_CASE(4)
                PL_M2_CASE(5)
                PL_M2_CASE(6)
            default:
                return hipErrorInvalidValue;
            }
#undef PL_M2_CASE
            return hipGetLastError();
        }
    }
    if constexpr (E == EST_HOM) {
        if (streaming && a.shadow16) { // homography pre-filter on the matrix cores (PG = 2 P groups of 32 points per chunk)
            const dim3 mgrid(xcd_slices(slices * (uint32_t)kScoreThreads / (uint32_t)kMfmaThreads), chunks);
            const dim3 mblock(kMfmaThreads);
#define PL_MH_CASE(PP)                                                                                                 \
    case PP:                                                                                                           \
        k_score_mfmah<2 * PP><<<mgrid, mblock, 0, stream>>>(a.pts, static_cast<const uint4 *>(a.shadow16), a.models,    \
                                                          a.slots, a.num_hyp, a.hyp_capacity, a.thr2, pf, a.part_count, \
                                                          a.part_score);                                                \
        break;
            switch (P) {
                PL_MH_CASE(1)
                PL_MH_CASE(2)
                PL_MH_CASE(3)
                PL_MH_CASE(4)
                PL_MH_CASE(5)
            default:
                return hipErrorInvalidValue;
            }
#undef PL_MH_CASE
            return hipGetLastError();
        }
    }
    if constexpr (E == EST_ABS) {
        if (streaming && a.shadow16) { // pre-filter on the matrix cores (PG = 2 P groups of 32 points per wave)
            // two workgroups fit a CU (LDS, registers): one resident round of 512 workgroups, as for the Sampson form above
            const uint32_t mslices = std::min<uint32_t>(slices * (uint32_t)kScoreThreads / (uint32_t)kMfmaThreads,
                                                        std::max<uint32_t>(1u, 512u / chunks));
            const dim3 mgrid(xcd_slices(mslices), chunks);
            const dim3 mblock(kMfmaThreads);
#define PL_M_CASE(PP)                                                                                                  \
    case PP:                                                                                                           \
        k_score_mfma<2 * PP><<<mgrid, mblock, 0, stream>>>(a.pts, static_cast<const uint4 *>(a.shadow16), a.models,      \
                                                         a.slots, a.num_hyp, a.hyp_capacity, a.thr2, pf,               \
                                                         a.part_count, a.part_score, a.tickets);                       \
        break;
            switch (P) {
                PL_M_CASE(1)
                PL_M_CASE(2)
                PL_M_CASE(3)
                PL_M_CASE(4)
                PL_M_CASE(5)
            default: // (PG = 12 does not fit three workgroups into a CU's LDS - 4 instead of 6 wavefronts per SIMD: not built)
                return hipErrorInvalidValue;
            }
#undef PL_M_CASE
            return hipGetLastError();
        }
    }
    if (streaming) {
        const dim3 qgrid(std::max<uint32_t>(1u, slices * (uint32_t)kScoreThreads / (uint32_t)kQueueThreads), chunks);
        const dim3 qblock(kQueueThreads);
#define PL_Q_CASE(PP)                                                                                                  \
    case PP:                                                                                                           \
        k_score_queue<E, PP><<<qgrid, qblock, 0, stream>>>(a.pts, a.shadow, a.compact64, a.num_hyp, a.hyp_capacity,      \
                                                         a.thr2, pf, a.part_count, a.part_score, a.tickets);           \
        break;
        switch (P) {
            PL_Q_CASE(1)
            PL_Q_CASE(2)
            PL_Q_CASE(3)
            PL_Q_CASE(4)
            PL_Q_CASE(5)
            PL_Q_CASE(6)
        default:
            return hipErrorInvalidValue;
        }
#undef PL_Q_CASE
        return hipGetLastError();
    }
    return hipErrorInvalidValue; // only the streaming form exists (models decisions are taken on: k_score_seq)
}

hipError_t launch_score_seq(int est, const SeqScoreArgs &a, hipStream_t stream) {
    if (a.cap == 0)
        return hipSuccess;
    const dim3 grid(std::min<uint32_t>(a.cap, 128u)), block(kSeqThreads); // (candidate lists hold a few dozen entries)
    PL_DISPATCH_EST(est, k_score_seq<E><<<grid, block, 0, stream>>>(a));
    return hipGetLastError();
}
hipError_t launch_score(int est, const ScoreArgs &a, uint32_t slices, hipStream_t stream) {
    PL_DISPATCH_EST(est, return launch_score_est<E>(a, slices, stream));
    return hipSuccess;
}


hipError_t launch_lm(int est, const PointSet &pts, LMTask *tasks, uint32_t num_tasks, hipStream_t stream) {
    // (every task carries its correspondences: LMTask.pts; here they all are this problem's)
    const size_t want = sizeof(double) * point_doubles(est) * (size_t)pts.n;
    return launch_lm_tasks(est, tasks, num_tasks, want <= 128 * 1024 ? pts.n : 0u, stream);
}
// 0 (default): k_lm - tree sums beyond 256 correspondences - for poses and homographies, k_lm_ordered for FUNDAMENTAL
//    matrices: every refinement of an F enters through an SVD whose third singular value is at rounding level, and the
//    sign of the F that comes back hangs on that value's sign (pl_svd3.h) - the refined F that feeds the next refinement
//    (local optimisation -> final refinement of the loop -> the front-end's bundle) has to carry the reference's BITS,
//    not only its value to 1e-13, or the caller gets -F in half of the runs;
// 1: k_lm_ordered for every estimator (every sum in the reference's order at every n);
// 2: k_lm for every estimator (fastest; the sign of F is then unpinned above 256 correspondences);
// -1: not set yet (POSELIB_AMD_LM_ORDERED = 0 / 1 / 2 decides at the first launch).  pl_set_lm_mode() in the C-ABI.
static std::atomic<int> g_lm_mode{-1};
void set_lm_mode(int mode) { g_lm_mode.store((mode == 1 || mode == 2) ? mode : 0, std::memory_order_release); }
int get_lm_mode() {
    int m = g_lm_mode.load(std::memory_order_acquire);
    if (m < 0) {
        const char *e = std::getenv("POSELIB_AMD_LM_ORDERED");
        m = (e && e[0] == '1') ? 1 : (e && e[0] == '2') ? 2 : 0;
        g_lm_mode.store(m, std::memory_order_release);
    }
    return m;
}
// (a calling thread can ask for the reference's order for the launches it makes itself: the driver re-runs a homography problem
// that way when a decision of its loop hung on the last bits of two refined models of opposite sign - driver.cc ransac_core)
static thread_local int tl_lm_force_ordered = 0;
void set_lm_force_ordered(int on) { tl_lm_force_ordered = on; }
bool lm_sums_ordered(int est) {
    if (tl_lm_force_ordered)
        return true;
    const int m = get_lm_mode();
    return m == 1 || (m == 0 && est == EST_FUND);
}
#ifdef PL_LM_PROFILE
extern "C" int pl_debug_lm_profile(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lm_prof), sizeof(unsigned long long) * 16) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lm_prof), z, sizeof(z)) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif
hipError_t launch_lm_tasks(int est, LMTask *tasks, uint32_t num_tasks, uint32_t max_points, hipStream_t stream) {
    if (num_tasks == 0)
        return hipSuccess;
    if (est < 0 || est > 3)
        return hipErrorInvalidValue;
    const int ordered = lm_sums_ordered(est) ? 1 : 0;
    // stage the points in LDS when they fit next to the kernel's static LDS (160 KB per CU, one workgroup per CU); tasks
    // of a mixed launch whose points do not fit the launch's dynamic LDS read them from L2
    // (a request the points do not fit into would only keep every other workgroup off the CU: no staging then)
    static std::atomic<int> dyn_limit[2][4] = {{{-1}, {-1}, {-1}, {-1}}, {{-1}, {-1}, {-1}, {-1}}}; // bytes of dynamic LDS the kernel may ask for
    int limit = dyn_limit[ordered][est].load(std::memory_order_acquire);
    if (limit < 0) {
        hipFuncAttributes fa;
        hipError_t e = hipSuccess;
        if (ordered) {
            PL_DISPATCH_EST(est, e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_lm_ordered<E>)));
        } else {
            PL_DISPATCH_EST(est, e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_lm<E>)));
        }
        if (e != hipSuccess)
            return e;
        limit = std::max<int>(0, 160 * 1024 - (int)fa.sharedSizeBytes - 1024);
        limit = std::min<int>(limit, 128 * 1024);
        if (limit > 48 * 1024) {
            if (ordered) {
                PL_DISPATCH_EST(est, e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lm_ordered<E>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, limit));
            } else {
                PL_DISPATCH_EST(est, e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lm<E>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, limit));
            }
            if (e != hipSuccess)
                return e;
        }
        dyn_limit[ordered][est].store(limit, std::memory_order_release);
    }
    const size_t want = sizeof(double) * point_doubles(est) * (size_t)max_points;
    const bool lds = std::getenv("POSELIB_AMD_LM_NO_LDS") == nullptr;
    const size_t bytes = (lds && want <= (size_t)limit) ? want : 0;
    if (ordered) {
        PL_DISPATCH_EST(est, k_lm_ordered<E><<<dim3(num_tasks), dim3(kLMThreads), bytes, stream>>>(tasks, (uint32_t)bytes));
    } else {
        PL_DISPATCH_EST(est, k_lm<E><<<dim3(num_tasks), dim3(kLMThreads), bytes, stream>>>(tasks, (uint32_t)bytes));
    }
    return hipGetLastError();
}

__global__ void k_task_records(int est, const LMTask *tasks, const double *records_in, double *records_out,
                               uint32_t num_tasks, LMTask *host_tasks) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_tasks)
        return;
    if (host_tasks) { // the outputs of the LM kernel, straight to pinned host memory
        for (int i = 0; i < kParamDoubles; ++i)
            host_tasks[j].params[i] = tasks[j].params[i];
        host_tasks[j].iterations = tasks[j].iterations;
        host_tasks[j].skipped = tasks[j].skipped;
        host_tasks[j].cost = tasks[j].cost;
        host_tasks[j].initial_cost = tasks[j].initial_cost;
    }
    double *out = records_out + (size_t)j * kModelStride;
    if (tasks[j].skipped) { // refinement not run: model unchanged (relative_pose.cc:75-77)
        for (int i = 0; i < kModelStride; ++i)
            out[i] = records_in[(size_t)j * kModelStride + i];
    } else {
        record_from_lm_params(est, tasks[j].params, out);
    }
}
__global__ void k_select_record(const double *score_refined, double incumbent_score, const double *rec_refined,
                                const double *rec_incumbent, double *out) {
    const double *src = (*score_refined < incumbent_score) ? rec_refined : rec_incumbent;
    if (threadIdx.x < kModelStride)
        out[threadIdx.x] = src[threadIdx.x];
}
__global__ void k_select_record_g(const SelectArgs *arr) {
    const SelectArgs &a = arr[blockIdx.z];
    if (!a.out)
        return;
    const double *src = (*a.score_refined < a.incumbent_score) ? a.rec_refined : a.rec_incumbent;
    if (threadIdx.x < kModelStride)
        a.out[threadIdx.x] = src[threadIdx.x];
    if (threadIdx.x == 63 && a.fetch_src)
        *a.fetch_dst = *a.fetch_src;
    if (threadIdx.x == 62 && a.count_out)
        *a.count_out = (*a.score_refined < a.incumbent_score) ? *a.count_refined : a.count_incumbent;
}
hipError_t launch_group_select(const SelectArgs *args, uint32_t G, hipStream_t stream) {
    if (G == 0)
        return hipSuccess;
    k_select_record_g<<<dim3(1, 1, G), dim3(64), 0, stream>>>(args);
    return hipGetLastError();
}
hipError_t launch_group_mask(int est, const MaskArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream) {
    if (G == 0 || max_n == 0)
        return hipSuccess;
    const dim3 grid((max_n + 255) / 256, 1, G), block(256);
    PL_DISPATCH_EST(est, k_mask_g<E><<<grid, block, 0, stream>>>(args));
    return hipGetLastError();
}
hipError_t launch_group_score_seq(int est, const SeqScoreArgs *args, uint32_t G, uint32_t max_cap, hipStream_t stream) {
    if (G == 0 || max_cap == 0)
        return hipSuccess;
    const dim3 grid(std::min<uint32_t>(max_cap, 128u), 1, G), block(kSeqThreads);
    PL_DISPATCH_EST(est, k_score_seq_g<E><<<grid, block, 0, stream>>>(args));
    return hipGetLastError();
}

// ---- one batch step of a whole group of problems ----
int group_points_per_lane(int est) { return (est == EST_REL || est == EST_FUND) ? 6 : 5; }

template <int E> static hipError_t launch_group_score_est(const GroupArgs *args, const GroupDims &d, hipStream_t stream) {
    const dim3 grid(std::max<uint32_t>(1u, d.max_slices), std::max<uint32_t>(1u, d.max_chunks), d.G);
    if constexpr (E == EST_ABS) {
        if (d.any_mfma)
            k_score_mfma_g<10><<<grid, dim3(kMfmaThreads), 0, stream>>>(args);
        if (d.any_queue)
            k_score_queue_g<EST_ABS, 5><<<grid, dim3(kQueueThreads), 0, stream>>>(args);
    } else if constexpr (E == EST_HOM) {
        if (d.any_mfma)
            k_score_mfmah_g<10><<<grid, dim3(kMfmaThreads), 0, stream>>>(args);
        if (d.any_queue)
            k_score_queue_g<EST_HOM, 5><<<grid, dim3(kQueueThreads), 0, stream>>>(args);
    } else {
        // (problems on the matrix-core form cut their correspondences into chunks of 32 PG, the others into 64 * 6: the
        // grid covers the larger chunk count, every block checks its own problem's)
        if (d.any_mfma)
            k_score_mfma2_g<E, 2 * group_mfma2_points_per_lane(E)><<<grid, dim3(kMfmaThreads), 0, stream>>>(args);
        if (d.any_queue)
            k_score_queue_g<E, 6><<<grid, dim3(kQueueThreads), 0, stream>>>(args);
    }
    return hipGetLastError();
}

hipError_t launch_group_batch(int est, const GroupArgs *args, const GroupDims &d, hipStream_t stream, hipEvent_t ev0,
                              hipEvent_t ev1) {
    if (d.G == 0)
        return hipSuccess;
    hipError_t e = launch_group_positions(sample_size(est), args, d, stream);
    if (e != hipSuccess)
        return e;
    const dim3 ggrid((d.max_B + 63) / 64, 1, d.G), gblock(64);
    if (est == EST_REL) {
        e = launch_group_generate_rel(args, d.max_B, d.G, stream);
        if (e != hipSuccess)
            return e;
    } else {
        PL_DISPATCH_EST(est, k_generate_g<E><<<ggrid, gblock, 0, stream>>>(args));
    }
    e = launch_group_compact(args, d, stream);
    if (e != hipSuccess)
        return e;
    if (ev0 && (e = hipEventRecord(ev0, stream)) != hipSuccess)
        return e;
    PL_DISPATCH_EST(est, e = launch_group_score_est<E>(args, d, stream));
    if (e != hipSuccess)
        return e;
    if (ev1 && (e = hipEventRecord(ev1, stream)) != hipSuccess)
        return e;
    e = launch_group_finalize_records(args, d, stream);
    if (e != hipSuccess)
        return e;
    // candidates per problem: a handful (improving hypotheses of the batch; more than kRecordFirst = 64 sends the problem to
    // the single-problem path), every workgroup loops over the list with the grid's stride.  With many problems in the group
    // fewer workgroups per problem: the empty ones still cost a dispatch of 16 wavefronts each (batch_mixed, groups of 128:
    // 16 384 workgroups per launch, a third of the device time of that workload).
    const uint32_t per_problem = std::max<uint32_t>(16u, std::min<uint32_t>(128u, 4096u / std::max<uint32_t>(1u, d.G)));
    const dim3 sgrid(per_problem, 1, d.G), sblock(kSeqThreads);
    PL_DISPATCH_EST(est, k_score_seq_gb<E><<<sgrid, sblock, 0, stream>>>(args));
    return hipGetLastError();
}

hipError_t launch_task_records(int est, const LMTask *tasks, const double *records_in, double *records_out,
                               uint32_t num_tasks, LMTask *host_tasks, hipStream_t stream) {
    if (num_tasks == 0)
        return hipSuccess;
    k_task_records<<<dim3((num_tasks + 63) / 64), dim3(64), 0, stream>>>(est, tasks, records_in, records_out, num_tasks,
                                                                         host_tasks);
    return hipGetLastError();
}
hipError_t launch_select_record(const double *score_refined, double incumbent_score, const double *rec_refined,
                                const double *rec_incumbent, double *out, hipStream_t stream) {
    k_select_record<<<dim3(1), dim3(64), 0, stream>>>(score_refined, incumbent_score, rec_refined, rec_incumbent, out);
    return hipGetLastError();
}

// Multi-workgroup LM (k_lm2): `slices` workgroups per task, max_iterations + 2 launches.
size_t lm2_state_bytes(uint32_t num_tasks) { return sizeof(LM2State) * 2 * (size_t)num_tasks; }
size_t lm2_partial_bytes(uint32_t num_tasks, uint32_t slices) { return sizeof(double) * 2 * (size_t)num_tasks * slices * 48; }
hipError_t launch_lm2(int est, const PointSet &pts, LMTask *tasks, uint32_t num_tasks, uint32_t slices,
                      uint32_t max_iterations, void *states, double *partials, hipStream_t stream) {
    if (num_tasks == 0)
        return hipSuccess;
    if (est == EST_REL)
        return hipErrorInvalidValue; // the relative-pose LO (pre-filter, tangent basis) stays on k_lm
    const dim3 grid(slices, num_tasks), block(kLM2Threads);
    const uint32_t launches = max_iterations + 2;
    for (uint32_t l = 0; l < launches; ++l) {
        switch (est) {
        case EST_ABS:
            k_lm2<EST_ABS><<<grid, block, 0, stream>>>(pts, tasks, static_cast<LM2State *>(states), partials, (int)(l & 1), l == 0);
            break;
        case EST_FUND:
            k_lm2<EST_FUND><<<grid, block, 0, stream>>>(pts, tasks, static_cast<LM2State *>(states), partials, (int)(l & 1), l == 0);
            break;
        default:
            k_lm2<EST_HOM><<<grid, block, 0, stream>>>(pts, tasks, static_cast<LM2State *>(states), partials, (int)(l & 1), l == 0);
            break;
        }
    }
    return hipGetLastError();
}

hipError_t launch_mask(int est, const PointSet &pts, const double *model, double thr2, uint8_t *mask,
                       uint8_t *host_mask, hipStream_t stream) {
    if (pts.n == 0)
        return hipSuccess;
    const dim3 grid((pts.n + 255) / 256), block(256);
    PL_DISPATCH_EST(est, k_mask<E><<<grid, block, 0, stream>>>(pts, model, thr2, mask, host_mask));
    return hipGetLastError();
}

} // namespace pl
