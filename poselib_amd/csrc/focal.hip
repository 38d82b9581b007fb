// poselib_amd - kernels of the focal-length estimator (ransac_pnpf: robust/ransac.cc:58-75, FocalAbsolutePoseEstimator).
//
//   k_focal_setup      one lane = one RANSAC iteration: draw the sample of four correspondences from the iteration's position in
//                      the splitmix64 stream (or take the host's PROSAC sample), the null space of P3.5Pf's linear constraints
//                      (pl_solver_p35pf.h, step 1) -> N (12 x 5) and the scale of the image points into a workspace in HBM
//   k_focal_solve      one WAVEFRONT = one iteration, the matrices in LDS: the 235 coefficients (four per lane), the 25 x 35
//                      template, LU with partial pivoting with a column per lane, the 10 x 10 action matrix, its eigenvalues by
//                      the lanes together (pl_eigen_wave.h), one lane per real eigenvalue for the 5 x 4 least-squares system and
//                      the pose; keeps the solutions the estimator keeps (focal >= 0, focal <= max_focal_length:
//                      absolute_pose.cc:89-95) in the order of the eigenvalues
//   k_focal_score      one wavefront = one model: compute_msac_score(Image, ...) (utils.cc:66-98) - inlier count and the sum of
//                      the inliers' squared residuals IN CORRESPONDENCE ORDER (the score decides comparisons in the loop, so
//                      it has to be the sequential sum): the lanes evaluate 64 correspondences at a time, the inliers' residuals
//                      are then added one by one in lane order (wave-uniform loop over the ballot).
//   k_focal_score_wg   the same score by one workgroup per model (producers / ordered chain): the few refined models of a local
//                      optimisation
//   k_focal_mask       get_inliers(Image, ...) (utils.cc:385-399), one thread per correspondence.
// Rounds 3 - 5 solved P3.5Pf in a formulation of this project's own (29 x 35 Gauss-Jordan elimination in registers, packed
// eigenvalue / null-vector kernels for large launches; CHANGELOG.md); round 6 restates the reference's action-matrix template so
// that the solver returns the reference's roots bit for bit ("The focal-length estimators").
#include "pl_focal.h"
#include "pl_kernels.h"
#include "pl_solver_p35pf.h"
#include "pl_eigen_wave.h"
#include "pl_lm_chain.inc"
#include <algorithm>
#include <atomic>

namespace pl {

namespace {

// ---- the generator: two kernels over a workspace in HBM (element e of sample it at stage[e * B + it]) ------------------------
constexpr int kStageN = 0, kStageF0 = 60, kStageDoubles = 61;

// (the kernels' bodies are functions of (arguments, block index): the single-problem kernels pass their own argument block, the
// group kernels - blockIdx.y = member of the group - the member's entry of a device-resident table, read before any store)
__device__ __forceinline__ void focal_setup_body(const FocalGenArgs &g, uint32_t blk) {
    const uint32_t it = blk * 64 + threadIdx.x;
    if (it >= g.num_iters)
        return;
    double xs[8];
    Vec3 X[4];
    if (g.explicit_in) { // minimal problems given explicitly: [x 4 x 2 | X 4 x 3]
        const double *p = g.explicit_in + (size_t)it * 20;
        for (int k = 0; k < 8; ++k)
            xs[k] = p[k];
        for (int k = 0; k < 4; ++k)
            X[k] = v3(p[8 + 3 * k], p[9 + 3 * k], p[10 + 3 * k]);
    } else {
        uint32_t idx[kFocalSample];
        if (g.samples) { // PROSAC: drawn on the host
            for (int k = 0; k < kFocalSample; ++k)
                idx[k] = g.samples[(size_t)it * kFocalSample + k];
        } else {
            draw_sample<kFocalSample>(g.seed, g.pos_base + g.positions[it], g.n, idx);
        }
        for (int k = 0; k < 4; ++k) {
            xs[2 * k] = g.a[0][idx[k]];
            xs[2 * k + 1] = g.a[1][idx[k]];
            X[k] = v3(g.a[2][idx[k]], g.a[3][idx[k]], g.a[4][idx[k]]);
        }
    }
    const size_t B = g.num_iters;
    double N[60], f0;
    p35pf_nullspace(xs, X, N, f0);
    for (int e = 0; e < 60; ++e)
        g.stage[(size_t)(kStageN + e) * B + it] = N[e];
    g.stage[(size_t)kStageF0 * B + it] = f0;
}
__global__ __launch_bounds__(64) void k_focal_setup(FocalGenArgs g) { focal_setup_body(g, blockIdx.x); }
__global__ __launch_bounds__(64) void k_focal_setup_g(const FocalGenArgs *__restrict__ gs) {
    const FocalGenArgs g = gs[blockIdx.y];
    focal_setup_body(g, blockIdx.x);
}

// k_focal_solve: one WAVEFRONT = one sample.  LDS of a wavefront (doubles):
//   [0, 60) N | [60, 296) the coefficients, later the action matrix (100) | [296, 296 + 25 * 36) the template, row-major at a row
//   stride of 36 (consecutive lanes = consecutive columns: no bank conflicts), later the eigenvalue workspace (130)
constexpr int kSolveWaves = 4;
constexpr int kLuStride = 36, kLdsN = 0, kLdsCoef = 60, kLdsC = 296, kSolveLds = kLdsC + 25 * kLuStride;
static_assert(kP35Coeffs <= kLdsC - kLdsCoef && kP35Cols <= kLuStride, "regions");
static_assert(eig_wave_doubles(10) <= 25 * kLuStride, "the eigenvalue workspace takes the template's place");
__device__ __forceinline__ void focal_solve_body(const FocalGenArgs &g, uint32_t blk) {
    __shared__ double s_solve[kSolveWaves][kSolveLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    const size_t B = g.num_iters;
    double *base = s_solve[wave], *Ns = base + kLdsN, *coef = base + kLdsCoef, *C = base + kLdsC;
    if (lane < 60)
        Ns[lane] = g.stage[(size_t)(kStageN + lane) * B + it];
    const double f0 = g.stage[(size_t)kStageF0 * B + it];
    PL_WAVE_SYNC();
    // ---- coefficients, template, the last five rows of C0^-1 C1 (p35pf.cc:86-871)
    template_coefficients_wave<false, kP35Coeffs>(Ns, kP35TermStart, kP35TermPacked, coef, lane);
    PL_WAVE_SYNC();
    template_fill_wave<25, kP35Cols, kLuStride>(coef, kP35ColStart, kP35EntryRow, kP35EntryCoeff, C, lane);
    lu_solve_tail_wave<25, kP35Cols, kLuStride, 5>(C, lane);
    // ---- the action matrix (p35pf.cc:873-885): kept in the coefficients' place for the roots, a copy for the eigenvalue iteration
    double am0, am1;
    {
        auto tail = [&](int r, int j) { return C[(20 + r) * kLuStride + 25 + j]; };
        am0 = p35pf_action_entry(tail, lane / 10, lane % 10);
        am1 = lane + 64 < 100 ? p35pf_action_entry(tail, (lane + 64) / 10, (lane + 64) % 10) : 0.0;
    }
    PL_WAVE_SYNC();
    double *am = coef, *eig = C;
    am[lane] = am0, eig[lane] = am0;
    if (lane + 64 < 100)
        am[lane + 64] = am1, eig[lane + 64] = am1;
    pl_general_eigenvalues_wave<10>(eig, lane);
    // ---- one lane per real eigenvalue (|imag| < 1e-6, in the routine's order: p35pf.cc:890-896), the estimator's filter
    const double *wr = eig + 100 + 10, *wi = wr + 10;
    const bool real = lane < 10 && fabs(wi[lane < 10 ? lane : 0]) < 1e-6;
    bool valid = false;
    P35Solution sol;
    if (real) {
        p35pf_root_solution((const double *)am, wr[lane], (const double *)Ns, f0, sol);
        valid = true;
        if (!g.keep_all) { // absolute_pose.cc:89-95
            if (sol.focal < 0)
                valid = false;
            if (g.max_focal >= 0 && sol.focal > g.max_focal)
                valid = false;
        }
    }
    const uint64_t mask = __builtin_amdgcn_ballot_w64(valid);
    if (valid) {
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        FocalModel o;
        o.q[0] = sol.q.w, o.q[1] = sol.q.x, o.q[2] = sol.q.y, o.q[3] = sol.q.z;
        o.t[0] = sol.t.x, o.t[1] = sol.t.y, o.t[2] = sol.t.z;
        o.f = sol.focal;
        g.models[(size_t)it * kFocalMaxModels + pos] = o;
        if (g.host_models)
            g.host_models[(size_t)it * kFocalMaxModels + pos] = o;
    }
    if (lane == 0) {
        const uint32_t m = (uint32_t)__popcll(mask);
        g.num_models[it] = m;
        if (g.host_num_models)
            g.host_num_models[it] = m;
    }
}
__global__ __launch_bounds__(64 * kSolveWaves) void k_focal_solve(FocalGenArgs g) { focal_solve_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kSolveWaves) void k_focal_solve_g(const FocalGenArgs *__restrict__ gs) {
    const FocalGenArgs g = gs[blockIdx.y];
    focal_solve_body(g, blockIdx.x);
}

constexpr int kFocalScoreThreads = 256;

__device__ __forceinline__ void focal_score_body(const FocalScoreArgs &a, uint32_t blk) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blk * (kFocalScoreThreads / 64) + (threadIdx.x >> 6);
    if (slot >= a.num_slots)
        return;
    if (a.num_models && (slot % kFocalMaxModels) >= a.num_models[slot / kFocalMaxModels]) { // (wave-uniform)
        if (lane == 0) { // (the host never reads garbage - and no fill dispatches in front of this kernel)
            a.counts[slot] = 0;
            a.sums[slot] = 0.0;
        }
        return;
    }
    FocalModel m;
    if (a.lm_tasks) { // the refined pose and focal length of k_lm_cam's task
        const LMTask &t = a.lm_tasks[slot];
        for (int i = 0; i < 4; ++i)
            m.q[i] = t.params[i];
        for (int i = 0; i < 3; ++i)
            m.t[i] = t.params[4 + i];
        m.f = t.cam.p[0];
    } else {
        m = a.models[slot];
    }
    double R[9];
    focal_rotation(m, R);
    uint32_t count = 0;
    double sum = 0.0;
    for (uint32_t base = 0; base < a.n; base += 64u) {
        const uint32_t i = base + lane;
        double r2 = 0.0;
        bool in = false;
        if (i < a.n)
            in = focal_reproj_inlier(R, m.t, m.f, a.a[0][i], a.a[1][i], a.a[2][i], a.a[3][i], a.a[4][i], a.thr2, r2);
        unsigned long long bits = __builtin_amdgcn_ballot_w64(in);
        count += (uint32_t)__builtin_popcountll(bits);
        while (bits) { // the inliers of the chunk, in correspondence order
            const int l = __builtin_ctzll(bits);
            bits &= bits - 1;
            sum += __shfl(r2, l, 64);
        }
    }
    if (lane == 0) {
        a.counts[slot] = count;
        a.sums[slot] = sum;
    }
}
__global__ __launch_bounds__(kFocalScoreThreads) void k_focal_score(FocalScoreArgs a) { focal_score_body(a, blockIdx.x); }
__global__ __launch_bounds__(kFocalScoreThreads) void k_focal_score_g(const FocalScoreArgs *__restrict__ as) {
    const FocalScoreArgs a = as[blockIdx.y];
    focal_score_body(a, blockIdx.x);
}

// The same score by ONE WORKGROUP per model (round 4): wavefronts 1 .. 3 evaluate rounds of 192 correspondences into one of two LDS
// buffers (the inliers' squared residuals, zeros for the others: x + 0.0 = x), lane 0 of wavefront 0 adds the previous round with the
// inline-asm chain of k_lm_ordered (pl_lm_chain.inc: 11.7 cycles per term) - a model with 1200 inliers takes 18 us instead of 65
// (one wavefront evaluating 64 correspondences at a time and adding its inliers through v_readlane), which is the length of the
// launches that score the few refined models of a local optimisation.
constexpr int kScoreProd = kFocalScoreThreads - 64;
__device__ __forceinline__ void focal_score_wg_body(const FocalScoreArgs &a, uint32_t slot) {
    __shared__ __attribute__((aligned(16))) double s_terms[2][kScoreProd];
    __shared__ uint32_t s_cnt[kFocalScoreThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (slot >= a.num_slots) // (uniform; group launches: the grid is the largest member's)
        return;
    if (a.num_models && (slot % kFocalMaxModels) >= a.num_models[slot / kFocalMaxModels]) { // (uniform)
        if (threadIdx.x == 0) { // (the host never reads garbage - and no fill dispatches in front of this kernel)
            a.counts[slot] = 0;
            a.sums[slot] = 0.0;
        }
        return;
    }
    FocalModel m;
    if (a.lm_tasks) { // the refined pose and focal length of k_lm_cam's task
        const LMTask &t = a.lm_tasks[slot];
        for (int i = 0; i < 4; ++i)
            m.q[i] = t.params[i];
        for (int i = 0; i < 3; ++i)
            m.t[i] = t.params[4 + i];
        m.f = t.cam.p[0];
    } else {
        m = a.models[slot];
    }
    double R[9];
    focal_rotation(m, R);
    const uint32_t rounds = (a.n + (uint32_t)kScoreProd - 1u) / (uint32_t)kScoreProd;
    uint32_t count = 0;
    double sum = 0.0;
    for (uint32_t r = 0; r <= rounds; ++r) {
        if (wave > 0) {
            if (r < rounds) {
                const uint32_t i = r * (uint32_t)kScoreProd + (uint32_t)((wave - 1) * 64 + lane);
                double r2 = 0.0;
                bool in = false;
                if (i < a.n)
                    in = focal_reproj_inlier(R, m.t, m.f, a.a[0][i], a.a[1][i], a.a[2][i], a.a[3][i], a.a[4][i], a.thr2, r2);
                s_terms[r & 1u][(wave - 1) * 64 + lane] = in ? r2 : 0.0;
                count += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in)); // (every lane holds its wavefront's count)
            }
        } else if (r > 0 && threadIdx.x == 0) {
#pragma unroll 1
            for (int q = 0; q < kScoreProd; q += 64) {
                const uint32_t addr = (uint32_t)(uintptr_t)&s_terms[(r - 1u) & 1u][q];
                PL_LM_CHAIN64(sum, addr);
            }
        }
        __syncthreads();
    }
    if (lane == 0)
        s_cnt[wave] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int w = 1; w < kFocalScoreThreads / 64; ++w)
            c += s_cnt[w];
        a.counts[slot] = c;
        a.sums[slot] = sum;
    }
}
__global__ __launch_bounds__(kFocalScoreThreads) void k_focal_score_wg(FocalScoreArgs a) { focal_score_wg_body(a, blockIdx.x); }
__global__ __launch_bounds__(kFocalScoreThreads) void k_focal_score_wg_g(const FocalScoreArgs *__restrict__ as) {
    const FocalScoreArgs a = as[blockIdx.y];
    focal_score_wg_body(a, blockIdx.x);
}

__device__ __forceinline__ void focal_mask_body(const double *x, const double *y, const double *X, const double *Y, const double *Z, uint32_t n,
                                                const FocalModel &m, double thr2, uint8_t *mask, uint8_t *host_mask, uint32_t i) {
    if (i >= n)
        return;
    double R[9];
    focal_rotation(m, R);
    const uint8_t v = focal_reproj_mask(R, m.t, m.f, x[i], y[i], X[i], Y[i], Z[i], thr2) ? 1 : 0;
    mask[i] = v;
    if (host_mask)
        host_mask[i] = v;
}
__global__ void k_focal_mask(const double *x, const double *y, const double *X, const double *Y, const double *Z, uint32_t n,
                             FocalModel m, double thr2, uint8_t *mask, uint8_t *host_mask) {
    focal_mask_body(x, y, X, Y, Z, n, m, thr2, mask, host_mask, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void k_focal_mask_g(const FocalMaskArgs *__restrict__ as) {
    const FocalMaskArgs a = as[blockIdx.y];
    focal_mask_body(a.a[0], a.a[1], a.a[2], a.a[3], a.a[4], a.n, a.model, a.thr2, a.mask, a.host_mask, blockIdx.x * blockDim.x + threadIdx.x);
}

} // namespace

size_t focal_stage_bytes(uint32_t num_iters) { return sizeof(double) * (size_t)kStageDoubles * num_iters; }

hipError_t launch_focal_generate(const FocalGenArgs &g, hipStream_t stream) {
    if (g.num_iters == 0)
        return hipSuccess;
    if (!g.stage)
        return hipErrorInvalidValue;
    k_focal_setup<<<dim3((g.num_iters + 63u) / 64u), dim3(64), 0, stream>>>(g);
    k_focal_solve<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
    return hipGetLastError();
}
// ---- group launches (driver_focal_group.inc): blockIdx.y = member, the grid's x extent = the largest member's; `args` is a
// device-resident table of G entries.  Same bodies as the single-problem kernels: a member's results do not depend on its group.
hipError_t launch_focal_generate_g(const FocalGenArgs *args, uint32_t G, uint32_t max_iters, hipStream_t stream) {
    if (G == 0 || max_iters == 0)
        return hipSuccess;
    k_focal_setup_g<<<dim3((max_iters + 63u) / 64u, G), dim3(64), 0, stream>>>(args);
    k_focal_solve_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
    return hipGetLastError();
}
hipError_t launch_focal_score_g(const FocalScoreArgs *args, uint32_t G, uint32_t max_slots, bool workgroup_per_model, hipStream_t stream) {
    if (G == 0 || max_slots == 0)
        return hipSuccess;
    if (workgroup_per_model) {
        k_focal_score_wg_g<<<dim3(max_slots, G), dim3(kFocalScoreThreads), 0, stream>>>(args);
        return hipGetLastError();
    }
    constexpr uint32_t per_block = kFocalScoreThreads / 64;
    k_focal_score_g<<<dim3((max_slots + per_block - 1) / per_block, G), dim3(kFocalScoreThreads), 0, stream>>>(args);
    return hipGetLastError();
}
hipError_t launch_focal_mask_g(const FocalMaskArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream) {
    if (G == 0 || max_n == 0)
        return hipSuccess;
    k_focal_mask_g<<<dim3((max_n + 255u) / 256u, G), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}
// minimal problems given explicitly (pl_p35pf, pl_solve_focal_batch): in = count x [x 4 x 2 | X 4 x 3]; every solution is kept.
// stage: focal_stage_bytes(stage_samples) bytes; the problems go through it stage_samples at a time.
hipError_t launch_focal_solve(const double *in, uint32_t count, FocalModel *models, uint32_t *num_models, double *stage,
                              uint32_t stage_samples, hipStream_t stream) {
    if (stage_samples == 0)
        return hipErrorInvalidValue;
    for (uint32_t first = 0; first < count; first += stage_samples) {
        FocalGenArgs g{};
        g.explicit_in = in + (size_t)first * 20;
        g.num_iters = std::min(stage_samples, count - first);
        g.keep_all = 1;
        g.max_focal = -1.0;
        g.models = models + (size_t)first * kFocalMaxModels;
        g.num_models = num_models + first;
        g.stage = stage;
        hipError_t e = launch_focal_generate(g, stream);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}
hipError_t launch_focal_score(const FocalScoreArgs &a, hipStream_t stream) {
    if (a.num_slots == 0)
        return hipSuccess;
    if (a.num_slots <= 1024u) { // the refined models of a local optimisation: one workgroup per model
        k_focal_score_wg<<<dim3(a.num_slots), dim3(kFocalScoreThreads), 0, stream>>>(a);
        return hipGetLastError();
    }
    constexpr uint32_t per_block = kFocalScoreThreads / 64;
    k_focal_score<<<dim3((a.num_slots + per_block - 1) / per_block), dim3(kFocalScoreThreads), 0, stream>>>(a);
    return hipGetLastError();
}
hipError_t launch_focal_mask(const double *const *a, uint32_t n, const FocalModel &m, double thr2, uint8_t *mask, uint8_t *host_mask,
                             hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    k_focal_mask<<<dim3((n + 255u) / 256u), dim3(256), 0, stream>>>(a[0], a[1], a[2], a[3], a[4], n, m, thr2, mask, host_mask);
    return hipGetLastError();
}

} // namespace pl
