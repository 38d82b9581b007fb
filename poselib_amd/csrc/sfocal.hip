// poselib_amd - kernels of the shared-focal relative pose estimator (ransac_shared_focal_relpose: robust/ransac.cc:182-203,
// SharedFocalRelativePoseEstimator robust/estimators/relative_pose.cc:154-203, refiner robust/optim/relative.h:488-592).
//
//   k_sfocal_setup     one lane = one RANSAC iteration: the sample of six correspondences from the iteration's position in the
//                      splitmix64 stream (or the host's PROSAC sample), unit bearings, the null space of the epipolar constraints
//                      (pl_solver_6ptf.h, step 1) into a workspace in HBM
//   k_sfocal_solve     one WAVEFRONT = one iteration, the matrices in LDS: the 280 coefficients, the 31 x 46 template, LU with
//                      partial pivoting with a column per lane, the 15 x 15 action matrix, its characteristic polynomial
//                      (Danilevsky, a column per lane), the real roots (Sturm chain by one lane, the bisection level by level with a lane
//                      per live interval, Ridders / Newton with a lane per leaf: pl_sturm_n.h), one lane per
//                      root for the 7 x 7 system, then one lane per solution for the poses
//   k_sfocal_score     one wavefront = one model: compute_sampson_msac_score (utils.cc:204-239) of F = K_inv (E K_inv) - inlier
//                      count and the score IN CORRESPONDENCE ORDER (r2 of an inlier, the threshold of an outlier, one after the
//                      other: the score decides comparisons in the loop, so it has to be the sequential sum).  The lanes evaluate
//                      64 correspondences at a time; their terms are then added in lane order through v_readlane.
//   k_sfocal_score_wg  the same score by one workgroup per model (producers / ordered chain): the refined models of a local optimisation
//   k_sfocal_mask      get_inliers(F, ...) (utils.cc:401-419), one thread per correspondence.
//   k_sfocal_lm        one workgroup = one refinement, the whole Levenberg-Marquardt loop on the device: refine_model's
//                      pre-filter (Sampson error below 5 thr^2, nothing to do when <= 6 survive), then cost and normal equations
//                      summed correspondence after correspondence - wavefronts 1 .. 3 produce the entry terms of a round of 192
//                      correspondences, lane e < 27 of wavefront 0 adds entry e of [JtJ | Jtr] with the inline-asm chain - so that the
//                      refined model equals the oracle's to the bit for every n.
// Rounds 3 - 5 solved the six-point problem as a polynomial eigenvalue problem of this project's own (CHANGELOG.md); round 6 restates
// the reference's template (relpose_6pt_focal.cc) so that the solver returns the reference's roots.
#include "pl_kernels.h"
#include "pl_device.h"
#include "pl_sfocal.h"
#include "pl_solver_6ptf.h"
#include "pl_lm_chain.inc"
#include <algorithm>
#include <atomic>

namespace pl {

namespace {

// ---- the generator: two kernels over a workspace in HBM (element e of sample it at stage[e * B + it]) ------------------------
constexpr int kStNb = 0, kStX = 27, kStDoubles = 63;

// (the kernels' bodies are functions of (arguments, block index): the single-problem kernels pass their own argument block, the
// group kernels - blockIdx.y = member of the group - the member's entry of a device-resident table, read before any store)
__device__ __forceinline__ void sfocal_setup_body(const SFocalGenArgs &g, uint32_t blk) {
    const uint32_t it = blk * 64 + threadIdx.x;
    if (it >= g.num_iters)
        return;
    Vec3 x1[6], x2[6];
    if (g.explicit_in) { // minimal problems given explicitly: [x1 6 x 3 | x2 6 x 3] unit bearings
        const double *p = g.explicit_in + (size_t)it * 36;
        for (int k = 0; k < 6; ++k) {
            x1[k] = v3(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
            x2[k] = v3(p[18 + 3 * k], p[19 + 3 * k], p[20 + 3 * k]);
        }
    } else {
        uint32_t idx[kSFocalSample];
        if (g.samples) { // PROSAC: drawn on the host
            for (int k = 0; k < kSFocalSample; ++k)
                idx[k] = g.samples[(size_t)it * kSFocalSample + k];
        } else {
            draw_sample<kSFocalSample>(g.seed, g.pos_base + g.positions[it], g.n, idx);
        }
        for (int k = 0; k < 6; ++k) { // relative_pose.cc:157-160: homogeneous().normalized()
            x1[k] = bearing(g.a[0][idx[k]], g.a[1][idx[k]]);
            x2[k] = bearing(g.a[2][idx[k]], g.a[3][idx[k]]);
        }
    }
    const size_t B = g.num_iters;
    double *st = g.stage + it;
    double nb[27];
    six_nullspace(x1, x2, nb);
    for (int e = 0; e < 27; ++e)
        st[(size_t)(kStNb + e) * B] = nb[e];
    for (int k = 0; k < 6; ++k) {
        st[(size_t)(kStX + 3 * k) * B] = x1[k].x, st[(size_t)(kStX + 3 * k + 1) * B] = x1[k].y, st[(size_t)(kStX + 3 * k + 2) * B] = x1[k].z;
        st[(size_t)(kStX + 18 + 3 * k) * B] = x2[k].x, st[(size_t)(kStX + 19 + 3 * k) * B] = x2[k].y,
                                     st[(size_t)(kStX + 20 + 3 * k) * B] = x2[k].z;
    }
}
__global__ __launch_bounds__(64) void k_sfocal_setup(SFocalGenArgs g) { sfocal_setup_body(g, blockIdx.x); }
__global__ __launch_bounds__(64) void k_sfocal_setup_g(const SFocalGenArgs *__restrict__ gs) {
    const SFocalGenArgs g = gs[blockIdx.y];
    sfocal_setup_body(g, blockIdx.x);
}

constexpr int kSolveWaves = 4, kMaxRoots = 16;
// lane gl (< 16) = solution gl of sample `it`: essential matrix, up to four poses; the models leave in the order of the solutions
// (prefix sum of the counts).  x1 / x2 / nb: the sample's bearings and null space (LDS), cnt: 16 doubles of scratch (LDS).  Returns the
// number of models of the sample (every lane).
__device__ __forceinline__ uint32_t sfocal_emit_poses(const SFocalGenArgs &g, uint32_t it, int gl, int ns, double sxv, double syv, double swv,
                                                      const Vec3 *x1, const Vec3 *x2, const double *nb, double *cnt) {
    FocalModel mine[4];
    uint32_t c = 0;
    if (gl < ns) {
        // (bearings and null space are read from LDS where they are used: as local copies they cost 126 registers)
        six_solution_poses(x1, x2, nb, sxv, syv, swv, [&](Quat q, Vec3 t, double f) {
            FocalModel o;
            o.q[0] = q.w, o.q[1] = q.x, o.q[2] = q.y, o.q[3] = q.z;
            o.t[0] = t.x, o.t[1] = t.y, o.t[2] = t.z;
            o.f = f;
            if (c < 4u)
                mine[c] = o;
            ++c;
        });
    }
    if (gl < kMaxRoots)
        cnt[gl] = (double)c;
    PL_WAVE_SYNC();
    uint32_t off = 0, m = 0;
    for (int s = 0; s < ns; ++s) {
        const uint32_t cs = (uint32_t)cnt[s];
        off += s < gl ? cs : 0u;
        m += cs;
    }
    FocalModel *out = g.models + (size_t)it * kSFocalMaxModels;
    for (uint32_t i = 0; i < c && i < 4u; ++i) {
        out[off + i] = mine[i];
        if (g.host_models)
            g.host_models[(size_t)it * kSFocalMaxModels + off + i] = mine[i];
    }
    return m;
}

// k_sfocal_solve: one WAVEFRONT = one sample.  LDS of a wavefront (doubles):
//   [0, 28) null space | [28, 64) bearings | [64, 344) the coefficients, later the action matrix (225) |
//   [344, 344 + 31 * 46) the template, row-major (consecutive lanes = consecutive columns), later: the working copy of the action
//   matrix (225) | Danilevsky's vectors (30) | the polynomial (16) | roots (16) | solutions sx, sy, sw (48) | counts (16) | the Sturm
//   chain and the bisection's stack (175) | the leaves (60)
constexpr int kSixS = 46, kLdsNb = 0, kLdsX = 28, kLdsCoef = 64, kLdsC = 344, kSolveLds = kLdsC + 31 * kSixS;
constexpr int kLdsAmp = 0, kLdsWs = 225, kLdsPoly = 255, kLdsEv = 271, kLdsSol = 287, kLdsCnt = 335, kLdsSturm = 352,
              kLdsLeaves = kLdsSturm + kSturmNWork(15), kLdsLevel = kLdsLeaves + 2 * kSturmNLeaves(15); // (offsets inside the template's region)
static_assert(kSixCoeffs <= kLdsC - kLdsCoef && kLdsLevel + kSturmNWaveWork(15) <= 31 * kSixS, "regions");
__device__ __forceinline__ void sfocal_solve_body(const SFocalGenArgs &g, uint32_t blk) {
    __shared__ double s_fin[kSolveWaves][kSolveLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    const size_t B = g.num_iters;
    const double *st = g.stage + it;
    double *base = s_fin[wave], *nb = base + kLdsNb, *coef = base + kLdsCoef, *C = base + kLdsC;
    if (lane < 27)
        nb[lane] = st[(size_t)(kStNb + lane) * B];
    if (lane < 36)
        base[kLdsX + lane] = st[(size_t)(kStX + lane) * B];
    PL_WAVE_SYNC();
    // ---- coefficients, template, the last eight rows of C0^-1 C1 (relpose_6pt_focal.cc:54-1043)
#ifndef PL_SFOCAL_STOP
#define PL_SFOCAL_STOP 99 // (experiment builds: the kernel returns after phase n - profiles/r06_sfocal_phases.md)
#endif
#define PL_SFOCAL_PHASE(n)                                                                                             \
    if (PL_SFOCAL_STOP <= (n)) {                                                                                       \
        if (lane == 0)                                                                                                 \
            g.num_models[it] = 0;                                                                                      \
        return;                                                                                                        \
    }
    PL_SFOCAL_PHASE(0)
    template_coefficients_wave<true, kSixCoeffs>(nb, kSixTermStart, kSixTermPacked, coef, lane);
    PL_WAVE_SYNC();
    PL_SFOCAL_PHASE(1)
    template_fill_wave<31, kSixCols, kSixS>(coef, kSixColStart, kSixEntryRow, kSixEntryCoeff, C, lane);
    PL_SFOCAL_PHASE(2)
    lu_solve_tail_wave<31, kSixCols, kSixS, 8>(C, lane);
    PL_SFOCAL_PHASE(3)
    // ---- the action matrix (:1045-1053): kept in the coefficients' place for the roots, a working copy for the polynomial
    double amv[4];
    {
        auto tail = [&](int r, int j) { return C[(23 + r) * kSixS + 31 + j]; };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = lane + 64 * k;
            amv[k] = e < 225 ? six_action_entry(tail, e / 15, e % 15) : 0.0;
        }
    }
    PL_WAVE_SYNC();
    double *am = coef, *amp = C + kLdsAmp, *poly = C + kLdsPoly, *ev = C + kLdsEv, *sols = C + kLdsSol;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = lane + 64 * k;
        if (e < 225)
            am[e] = amv[k], amp[e] = amv[k];
    }
    PL_WAVE_SYNC();
    // ---- characteristic polynomial, its real roots (:1069-1076)
    PL_SFOCAL_PHASE(4)
    danilevsky_charpoly_wave<15>(amp, C + kLdsWs, poly, lane);
    PL_SFOCAL_PHASE(5)
    // the real roots: lane 0 builds the Sturm chain (its arrays in LDS: as private arrays they are scratch memory), the bisection
    // runs level by level with one lane per live interval (sturm_n_isolate_wave), then ONE LANE PER LEAF polishes (Ridders +
    // Newton) - the roots in leaf order, at most 15, as the serial routine emits them
    int nleaf = 0;
    unsigned tiny = 0;
    double *leaves = C + kLdsLeaves;
    nleaf = sturm_n_isolate_wave<15>(poly, 1e-12, C + kLdsSturm, leaves, tiny, C + kLdsLevel, lane);
    nleaf = __builtin_amdgcn_readfirstlane(nleaf);
    tiny = (unsigned)__builtin_amdgcn_readfirstlane((int)tiny);
    PL_WAVE_SYNC();
    double root = 0;
    bool has_root = false;
    if (lane < nleaf)
        has_root = sturm_n_leaf_root<15>(C + kLdsSturm, leaves[2 * lane], leaves[2 * lane + 1], (tiny >> lane) & 1u, 1e-12, &root) != 0;
    const uint64_t rmask = __builtin_amdgcn_ballot_w64(has_root);
    if (has_root) {
        const uint32_t rpos = __builtin_amdgcn_mbcnt_hi((uint32_t)(rmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rmask, 0u));
        if (rpos < 15u)
            ev[rpos] = root;
    }
    int nroots = (int)__popcll(rmask);
    nroots = nroots < 15 ? nroots : 15;
    PL_WAVE_SYNC();
    PL_SFOCAL_PHASE(6)
    // ---- lane s = root s: x and w (:11-52); w < 1e-8 dropped (:1105); the solutions in the order of the roots
    bool keep = false;
    double x = 0, w = 1;
    const double y = ev[lane < nroots ? lane : 0];
    if (lane < nroots) {
        six_root_xw((const double *)am, y, x, w);
        keep = !(w < 1e-8);
    }
    const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
    const int ns = (int)__popcll(mask);
    if (keep) {
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        sols[pos] = x, sols[kMaxRoots + pos] = y, sols[2 * kMaxRoots + pos] = w;
    }
    PL_WAVE_SYNC();
    PL_SFOCAL_PHASE(7)
    // ---- lane s = solution s: essential matrix, poses (:1107-1141)
    uint32_t m = 0;
    if (ns > 0) { // (uniform)
        const int s = lane < ns ? lane : 0;
        const Vec3 *x1 = reinterpret_cast<const Vec3 *>(base + kLdsX);
        m = sfocal_emit_poses(g, it, lane < kMaxRoots ? lane : kMaxRoots, ns, sols[s], sols[kMaxRoots + s], sols[2 * kMaxRoots + s], x1, x1 + 6, nb,
                              C + kLdsCnt);
    }
    if (lane == 0) {
        g.num_models[it] = m;
        if (g.host_num_models)
            g.host_num_models[it] = m;
    }
}
__global__ __launch_bounds__(64 * kSolveWaves) void k_sfocal_solve(SFocalGenArgs g) { sfocal_solve_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kSolveWaves) void k_sfocal_solve_g(const SFocalGenArgs *__restrict__ gs) {
    const SFocalGenArgs g = gs[blockIdx.y];
    sfocal_solve_body(g, blockIdx.x);
}

__device__ __forceinline__ double readlane_f64(double v, int l) { // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

constexpr int kSFocalScoreThreads = 256;

// the model of a score slot: models[slot], or the parameters k_sfocal_lm left in task `slot` (refined - or the seed it was given when
// refine_model returned without touching the model, relative_pose.cc:187-189)
__device__ __forceinline__ FocalModel sfocal_score_model(const SFocalScoreArgs &a, uint32_t slot) {
    if (!a.lm_tasks)
        return a.models[slot];
    const SFocalLMTask &t = a.lm_tasks[slot];
    FocalModel m;
    for (int i = 0; i < 4; ++i)
        m.q[i] = t.params[i];
    for (int i = 0; i < 3; ++i)
        m.t[i] = t.params[4 + i];
    m.f = t.params[kSFocalFocalSlot];
    return m;
}

// one wavefront = the model of slot `slot`
__device__ __forceinline__ void sfocal_score_slot(const SFocalScoreArgs &a, uint32_t slot, uint32_t lane) {
    const FocalModel m = sfocal_score_model(a, slot);
    double F[9];
    sfocal_F_score(m, F);
    uint32_t count = 0;
    double score = 0.0;
    for (uint32_t base = 0; base < a.n; base += 64u) {
        const uint32_t i = base + lane;
        double term = 0.0;
        bool in = false;
        if (i < a.n) {
            const double r2 = sampson_sq(F, a.a[0][i], a.a[1][i], a.a[2][i], a.a[3][i]);
            in = r2 < a.thr2;
            term = in ? r2 : a.thr2;
        }
        count += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
        const int valid = (int)min(64u, a.n - base);
        for (int l = 0; l < valid; ++l) // utils.cc:226-236, one correspondence after the other
            score += readlane_f64(term, l);
    }
    if (lane == 0) {
        a.counts[slot] = count;
        a.scores[slot] = score;
    }
}
// A batch of iterations (num_models given): one WORKGROUP = one iteration, its models over the workgroup's wavefronts - an iteration has
// 60 slots of which 0.2 hold a model, and a wavefront per slot was 3.8 M wavefronts per group launch to score 14 k models (0.75 ms, of
// which the scoring is half).  Models given one by one (num_models == nullptr): one wavefront per slot.
__device__ __forceinline__ void sfocal_score_body(const SFocalScoreArgs &a, uint32_t blk) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (a.num_models) {
        if ((size_t)blk * kSFocalMaxModels >= a.num_slots) // (group launches: the grid is the largest member's)
            return;
        const uint32_t nm = a.num_models[blk];
        for (uint32_t m = wave; m < nm; m += kSFocalScoreThreads / 64)
            sfocal_score_slot(a, blk * (uint32_t)kSFocalMaxModels + m, lane);
        return;
    }
    const uint32_t slot = blk * (kSFocalScoreThreads / 64) + wave;
    if (slot < a.num_slots)
        sfocal_score_slot(a, slot, lane);
}
__global__ __launch_bounds__(kSFocalScoreThreads) void k_sfocal_score(SFocalScoreArgs a) { sfocal_score_body(a, blockIdx.x); }
__global__ __launch_bounds__(kSFocalScoreThreads) void k_sfocal_score_g(const SFocalScoreArgs *__restrict__ as) {
    const SFocalScoreArgs a = as[blockIdx.y];
    sfocal_score_body(a, blockIdx.x);
}

// The same score by ONE WORKGROUP per model (round 4; used for the few refined models of a local optimisation - a batch of
// iterations has 60 slots per iteration of which 0.3 hold a model): wavefronts 1 .. 3 evaluate rounds of 192 correspondences into one
// of two LDS buffers, lane 0 of wavefront 0 adds the previous round's terms with the inline-asm chain of k_lm_ordered
// (pl_lm_chain.inc; zeros beyond n: x + 0.0 = x).
constexpr int kSfScoreProd = kSFocalScoreThreads - 64;
__device__ __forceinline__ void sfocal_score_wg_body(const SFocalScoreArgs &a, uint32_t slot) {
    __shared__ __attribute__((aligned(16))) double s_terms[2][kSfScoreProd];
    __shared__ uint32_t s_cnt[kSFocalScoreThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (slot >= a.num_slots) // (uniform; group launches: the grid is the largest member's)
        return;
    if (a.num_models && (slot % kSFocalMaxModels) >= a.num_models[slot / kSFocalMaxModels])
        return; // (uniform)
    const FocalModel m = sfocal_score_model(a, slot);
    double F[9];
    sfocal_F_score(m, F);
    const uint32_t rounds = (a.n + (uint32_t)kSfScoreProd - 1u) / (uint32_t)kSfScoreProd;
    uint32_t count = 0;
    double score = 0.0;
    for (uint32_t r = 0; r <= rounds; ++r) {
        if (wave > 0) {
            if (r < rounds) {
                const uint32_t i = r * (uint32_t)kSfScoreProd + (uint32_t)((wave - 1) * 64 + lane);
                double term = 0.0;
                bool in = false;
                if (i < a.n) {
                    const double r2 = sampson_sq(F, a.a[0][i], a.a[1][i], a.a[2][i], a.a[3][i]);
                    in = r2 < a.thr2;
                    term = in ? r2 : a.thr2;
                }
                s_terms[r & 1u][(wave - 1) * 64 + lane] = term;
                count += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in));
            }
        } else if (r > 0 && threadIdx.x == 0) {
#pragma unroll 1
            for (int q = 0; q < kSfScoreProd; q += 64) {
                const uint32_t addr = (uint32_t)(uintptr_t)&s_terms[(r - 1u) & 1u][q];
                PL_LM_CHAIN64(score, addr);
            }
        }
        __syncthreads();
    }
    if (lane == 0)
        s_cnt[wave] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int w = 1; w < kSFocalScoreThreads / 64; ++w)
            c += s_cnt[w];
        a.counts[slot] = c;
        a.scores[slot] = score;
    }
}
__global__ __launch_bounds__(kSFocalScoreThreads) void k_sfocal_score_wg(SFocalScoreArgs a) { sfocal_score_wg_body(a, blockIdx.x); }
__global__ __launch_bounds__(kSFocalScoreThreads) void k_sfocal_score_wg_g(const SFocalScoreArgs *__restrict__ as) {
    const SFocalScoreArgs a = as[blockIdx.y];
    sfocal_score_wg_body(a, blockIdx.x);
}

__device__ __forceinline__ void sfocal_mask_body(const double *x1, const double *y1, const double *x2, const double *y2, uint32_t n, const FocalModel &m,
                                                 double thr2, uint8_t *mask, uint8_t *host_mask, uint32_t i) {
    if (i >= n)
        return;
    double F[9];
    sfocal_F_score(m, F);
    const uint8_t v = sampson_sq(F, x1[i], y1[i], x2[i], y2[i]) < thr2 ? 1 : 0;
    mask[i] = v;
    if (host_mask)
        host_mask[i] = v;
}
__global__ void k_sfocal_mask(const double *x1, const double *y1, const double *x2, const double *y2, uint32_t n, FocalModel m,
                              double thr2, uint8_t *mask, uint8_t *host_mask) {
    sfocal_mask_body(x1, y1, x2, y2, n, m, thr2, mask, host_mask, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void k_sfocal_mask_g(const FocalMaskArgs *__restrict__ as) {
    const FocalMaskArgs a = as[blockIdx.y];
    sfocal_mask_body(a.a[0], a.a[1], a.a[2], a.a[3], a.n, a.model, a.thr2, a.mask, a.host_mask, blockIdx.x * blockDim.x + threadIdx.x);
}

constexpr int kSfLMThreads = 256;
constexpr int kSfLMWaves = kSfLMThreads / 64;
constexpr int kSfProd = kSfLMThreads - 64; // correspondences per round: wavefronts 1 .. 3 produce, wavefront 0 adds
constexpr int kSfTermStride = kSfProd + 2; // column stride of the entry terms (16-byte aligned columns)

__global__ __launch_bounds__(kSfLMThreads) void k_sfocal_lm(SFocalLMTask *tasks) {
    extern __shared__ __attribute__((aligned(16))) double s_rows[]; // two buffers of a round's entry terms: 2 x kSFocalEntries x kSfTermStride (cost pass: 2 x kSfProd terms)
    __shared__ SFocalLMTask s_task;
    __shared__ LMControl ctl;
    __shared__ double cur[kParamDoubles], trial[kParamDoubles];
    __shared__ SFocalCtx ctx;
    __shared__ double normal[kSFocalEntries];
    __shared__ uint32_t s_wcnt[kSfLMWaves];
    __shared__ uint32_t s_rcnt[2][kSfLMWaves]; // rows a producer wavefront left in its third of buffer 0 / 1
    __shared__ double s_racc;
    __shared__ uint32_t s_count;
    __shared__ double s_F[9];

    SFocalLMTask &Tout = tasks[blockIdx.x];
    {
        static_assert(sizeof(SFocalLMTask) % 8 == 0, "copied as 64-bit words");
        const uint64_t *src = reinterpret_cast<const uint64_t *>(&Tout);
        uint64_t *dst = reinterpret_cast<uint64_t *>(&s_task);
        for (uint32_t w = threadIdx.x; w < sizeof(SFocalLMTask) / 8; w += kSfLMThreads)
            dst[w] = src[w];
        __syncthreads();
    }
    const SFocalLMTask &T = s_task;
    const uint32_t n = T.n;
    const double *x1 = T.a[0], *y1 = T.a[1], *x2 = T.a[2], *y2 = T.a[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *mask = T.mask;

    auto block_count = [&](uint32_t mine) -> uint32_t { // sum over the workgroup, returned to every lane
        const uint32_t ws = wave_sum_u32(mine);
        __syncthreads();
        if (lane == 0)
            s_wcnt[wave] = ws;
        __syncthreads();
        uint32_t tot = 0;
        for (int w = 0; w < kSfLMWaves; ++w)
            tot += s_wcnt[w];
        return tot;
    };

    if (T.prefilter_thr2 > 0) { // relative_pose.cc:179-190
        if (threadIdx.x == 0) {
            FocalModel m;
            for (int i = 0; i < 4; ++i)
                m.q[i] = T.params[i];
            for (int i = 0; i < 3; ++i)
                m.t[i] = T.params[4 + i];
            m.f = T.params[kSFocalFocalSlot];
            sfocal_F_score(m, s_F);
        }
        __syncthreads();
        double F[9];
        for (int i = 0; i < 9; ++i)
            F[i] = s_F[i];
        uint32_t kept = 0;
        for (uint32_t i = threadIdx.x; i < n; i += kSfLMThreads) {
            const uint8_t v = sampson_sq(F, x1[i], y1[i], x2[i], y2[i]) < T.prefilter_thr2 ? 1 : 0;
            T.scratch[i] = v;
            kept += v;
        }
        const uint32_t total = block_count(kept);
        if (total <= 6) {
            if (threadIdx.x == 0) {
                Tout.iterations = 0;
                Tout.skipped = 1u;
                Tout.cost = Tout.initial_cost = 0.0;
            }
            return;
        }
        mask = T.scratch;
        __threadfence_block();
        __syncthreads();
    }

    if (threadIdx.x == 0) {
        for (int i = 0; i < kParamDoubles; ++i)
            cur[i] = T.params[i];
        ctl.opt = T.opt;
        ctl.loss = make_loss(T.opt.loss_type, T.opt.loss_scale);
        ctl.done = 0;
    }
    __syncthreads();

    // Both passes are a two-stage pipeline over rounds of kSfProd = 192 correspondences (round 4, like k_lm_cam; up to then every
    // wavefront produced a round of 256, waited, and watched wavefront 0 add): wavefronts 1 .. 3 evaluate round r into buffer r & 1
    // while wavefront 0 adds round r - 1 from the other buffer; ONE barrier per round.  The sums run over the correspondences one
    // after the other, as the reference adds them: each producer wavefront compacts the rows of ITS 64 correspondences into its own
    // third of the buffer (ballot + v_mbcnt), the consumer walks the thirds in order.
    const uint32_t rounds = (n + (uint32_t)kSfProd - 1u) / (uint32_t)kSfProd;
    const int pw = wave - 1; // producer wavefront 0 .. 2

    // robust cost at p -> s_racc, s_count: every correspondence's term (zeros for skipped ones: x + 0.0 = x), lane 0 adds a round's
    // 192 terms with the inline-asm chain of k_lm_ordered (pl_lm_chain.inc)
    auto cost_pass = [&](const double *p) {
        if (threadIdx.x == 0)
            sfocal_prepare(p, ctx, false);
        __syncthreads();
        const Loss loss = ctl.loss;
        double racc = 0.0; // (thread 0)
        uint32_t cnt = 0;
        for (uint32_t r = 0; r <= rounds; ++r) {
            if (wave > 0) {
                if (r < rounds) {
                    const uint32_t i = r * (uint32_t)kSfProd + (uint32_t)(pw * 64 + lane);
                    double term = 0.0;
                    const bool kept = i < n && !(mask && !mask[i]);
                    if (kept) {
                        const double res = sfocal_residual(ctx, x1[i], y1[i], x2[i], y2[i]);
                        term = 1.0 * loss_value(loss, res * res);
                    }
                    s_rows[(r & 1u) * kSfProd + pw * 64 + lane] = term;
                    cnt += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(kept)); // (every lane holds its wavefront's count)
                }
            } else if (r > 0 && threadIdx.x == 0) {
#pragma unroll 1
                for (int q = 0; q < kSfProd; q += 64) {
                    const uint32_t addr = (uint32_t)(uintptr_t)&s_rows[((r - 1u) & 1u) * kSfProd + q];
                    PL_LM_CHAIN64(racc, addr);
                }
            }
            __syncthreads();
        }
        if (lane == 0)
            s_wcnt[wave] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            s_racc = racc;
            uint32_t c = 0;
            for (int w = 1; w < kSfLMWaves; ++w)
                c += s_wcnt[w];
            s_count = c;
        }
        __syncthreads();
    };

    // normal equations at p -> normal[0 .. 27), s_count.  p's tangent basis is refreshed first (relative.h:513).
    // The PRODUCERS form the 27 entry terms of their correspondence (the products sfocal_entry_term forms, in its operand order)
    // and store them column-major, [entry][row]; every third is padded with zero rows to 64 (the lanes without a row write them:
    // x + 0.0 = x), so the consumer lane of an entry adds a third with ONE inline-asm chain (pl_lm_chain.inc: 11.7 cycles per row;
    // k_lm_cam's cycle counters showed the consumer of the row form - LDS reads and products per entry and row - as the bound).
    auto jacobian_pass = [&](double *p) {
        if (threadIdx.x == 0) {
            Refiner<EST_REL>::prepare_params(p);
            sfocal_prepare(p, ctx, true);
        }
        __syncthreads();
        const Loss loss = ctl.loss;
        double acc = 0.0;
        uint32_t total = 0; // (consumer)
        for (uint32_t r = 0; r <= rounds; ++r) {
            if (wave > 0) {
                if (r < rounds) {
                    const uint32_t i = r * (uint32_t)kSfProd + (uint32_t)(pw * 64 + lane);
                    double row[kSFocalRow];
                    bool kept = false;
                    if (i < n && !(mask && !mask[i]))
                        kept = sfocal_row(ctx, loss, x1[i], y1[i], x2[i], y2[i], row);
                    const uint64_t b = __builtin_amdgcn_ballot_w64(kept);
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                    const uint32_t rows = (uint32_t)__popcll(b);
                    if (lane == 0)
                        s_rcnt[r & 1u][wave] = rows;
                    // kept rows in ascending order at 0 .. rows - 1, zero rows behind them
                    const uint32_t pos = kept ? below : rows + ((uint32_t)lane - below);
                    double *dst = s_rows + (size_t)(r & 1u) * kSFocalEntries * kSfTermStride + (size_t)pw * 64 + pos;
                    if (kept) {
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int c = 0; c <= a; ++c, ++e) { // sfocal_entry_term, triangle entry (a, c)
                                const double t = row[2 + a] * row[2 + c];
                                dst[(size_t)e * kSfTermStride] = row[0] * t;
                            }
#pragma unroll
                        for (int k = 0; k < 6; ++k, ++e) { // gradient entry k
                            const double t = row[1] * row[2 + k];
                            dst[(size_t)e * kSfTermStride] = 1.0 * t;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < kSFocalEntries; ++e)
                            dst[(size_t)e * kSfTermStride] = 0.0;
                    }
                }
            } else if (r > 0) {
                const uint32_t buf = (r - 1u) & 1u;
                if (lane < kSFocalEntries) {
#pragma unroll 1
                    for (int w = 0; w < kSfLMWaves - 1; ++w) { // the three thirds in order
                        const uint32_t addr = (uint32_t)(uintptr_t)(s_rows + (size_t)buf * kSFocalEntries * kSfTermStride + (size_t)lane * kSfTermStride + (size_t)w * 64);
                        PL_LM_CHAIN64(acc, addr);
                    }
                }
                for (int w = 1; w < kSfLMWaves; ++w)
                    total += s_rcnt[buf][w];
            }
            __syncthreads(); // (the buffer of round r - 1 is rewritten by round r + 1)
        }
        if ((int)threadIdx.x < kSFocalEntries)
            normal[threadIdx.x] = acc;
        if (threadIdx.x == 0)
            s_count = total;
        __syncthreads();
    };

    cost_pass(cur);
    if (threadIdx.x == 0)
        lm_begin(ctl, T.opt, s_racc, s_count);
    __syncthreads();

    while (!ctl.done) {
        const bool fresh = ctl.rejac != 0;
        if (fresh)
            jacobian_pass(cur);
        if (threadIdx.x == 0) {
            lm_solve<6>(ctl, normal, fresh, s_count);
            if (!ctl.done)
                sfocal_step(cur, ctl.sol, trial);
        }
        __syncthreads();
        if (ctl.done)
            break;
        cost_pass(trial);
        if (threadIdx.x == 0) {
            if (lm_update<6>(ctl, normal, s_racc, s_count)) {
                for (int i = 0; i < kParamDoubles; ++i)
                    cur[i] = trial[i];
            }
        }
        __syncthreads();
    }

    if (threadIdx.x == 0) {
        for (int i = 0; i < kParamDoubles; ++i)
            Tout.params[i] = cur[i];
        Tout.iterations = ctl.iterations;
        Tout.skipped = 0u;
        Tout.cost = ctl.cost;
        Tout.initial_cost = ctl.initial_cost;
    }
}

} // namespace

size_t sfocal_stage_bytes(uint32_t num_iters) { return sizeof(double) * (size_t)kStDoubles * num_iters; }

hipError_t launch_sfocal_generate(const SFocalGenArgs &g, hipStream_t stream) {
    if (g.num_iters == 0)
        return hipSuccess;
    if (!g.stage)
        return hipErrorInvalidValue;
    k_sfocal_setup<<<dim3((g.num_iters + 63u) / 64u), dim3(64), 0, stream>>>(g);
    k_sfocal_solve<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
    return hipGetLastError();
}
// ---- group launches (driver_focal_group.inc): blockIdx.y = member, the grid's x extent = the largest member's; `args` is a
// device-resident table of G entries.  Same bodies as the single-problem kernels: a member's results do not depend on its group.
hipError_t launch_sfocal_generate_g(const SFocalGenArgs *args, uint32_t G, uint32_t max_iters, hipStream_t stream) {
    if (G == 0 || max_iters == 0)
        return hipSuccess;
    k_sfocal_setup_g<<<dim3((max_iters + 63u) / 64u, G), dim3(64), 0, stream>>>(args);
    k_sfocal_solve_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
    return hipGetLastError();
}
hipError_t launch_sfocal_score_g(const SFocalScoreArgs *args, uint32_t G, uint32_t max_slots, bool workgroup_per_model, hipStream_t stream) {
    if (G == 0 || max_slots == 0)
        return hipSuccess;
    if (workgroup_per_model) {
        k_sfocal_score_wg_g<<<dim3(max_slots, G), dim3(kSFocalScoreThreads), 0, stream>>>(args);
        return hipGetLastError();
    }
    // (a batch of iterations - every member's num_models is set: one workgroup per iteration)
    k_sfocal_score_g<<<dim3((max_slots + kSFocalMaxModels - 1) / kSFocalMaxModels, G), dim3(kSFocalScoreThreads), 0, stream>>>(args);
    return hipGetLastError();
}
hipError_t launch_sfocal_mask_g(const FocalMaskArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream) {
    if (G == 0 || max_n == 0)
        return hipSuccess;
    k_sfocal_mask_g<<<dim3((max_n + 255u) / 256u, G), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}
// minimal problems given explicitly (pl_relpose_6pt_shared_focal, pl_solve_focal_batch): in = count x [x1 6 x 3 | x2 6 x 3].
// stage: sfocal_stage_bytes(stage_samples) bytes; the problems go through it stage_samples at a time.
hipError_t launch_sfocal_solve(const double *in, uint32_t count, FocalModel *models, uint32_t *num_models, double *stage,
                               uint32_t stage_samples, hipStream_t stream) {
    if (stage_samples == 0)
        return hipErrorInvalidValue;
    for (uint32_t first = 0; first < count; first += stage_samples) {
        SFocalGenArgs g{};
        g.explicit_in = in + (size_t)first * 36;
        g.num_iters = std::min(stage_samples, count - first);
        g.models = models + (size_t)first * kSFocalMaxModels;
        g.num_models = num_models + first;
        g.stage = stage;
        hipError_t e = launch_sfocal_generate(g, stream);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}
hipError_t launch_sfocal_score(const SFocalScoreArgs &a, hipStream_t stream) {
    if (a.num_slots == 0)
        return hipSuccess;
    if (a.num_slots <= 1024u) { // the refined models of a local optimisation: one workgroup per model
        k_sfocal_score_wg<<<dim3(a.num_slots), dim3(kSFocalScoreThreads), 0, stream>>>(a);
        return hipGetLastError();
    }
    constexpr uint32_t per_block = kSFocalScoreThreads / 64;
    const uint32_t blocks = a.num_models ? (a.num_slots + kSFocalMaxModels - 1) / kSFocalMaxModels : (a.num_slots + per_block - 1) / per_block;
    k_sfocal_score<<<dim3(blocks), dim3(kSFocalScoreThreads), 0, stream>>>(a);
    return hipGetLastError();
}
hipError_t launch_sfocal_mask(const double *const *a, uint32_t n, const FocalModel &m, double thr2, uint8_t *mask, uint8_t *host_mask,
                              hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    k_sfocal_mask<<<dim3((n + 255u) / 256u), dim3(256), 0, stream>>>(a[0], a[1], a[2], a[3], n, m, thr2, mask, host_mask);
    return hipGetLastError();
}
hipError_t launch_sfocal_lm(SFocalLMTask *tasks, uint32_t num_tasks, hipStream_t stream) {
    if (num_tasks == 0)
        return hipSuccess;
    constexpr size_t bytes = sizeof(double) * 2 * kSFocalEntries * kSfTermStride; // 84 KB
    static std::atomic<int> prepared_dev[64]; // per device ordinal: the attribute is per-device state on some runtimes
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    std::atomic<int> &prepared = prepared_dev[dev_ & 63];
    if (!prepared.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sfocal_lm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess)
            return e;
        prepared.store(1, std::memory_order_release);
    }
    k_sfocal_lm<<<dim3(num_tasks), dim3(kSfLMThreads), bytes, stream>>>(tasks);
    return hipGetLastError();
}

} // namespace pl
