// poselib_amd - kernels of the focal-length estimator (ransac_pnpf: robust/ransac.cc:58-75, FocalAbsolutePoseEstimator).
//
//   k_focal_setup      one lane = one RANSAC iteration: draw the sample of four correspondences from the iteration's position in
//                      the splitmix64 stream (or take the host's PROSAC sample), null space and equations of P3.5Pf
//                      (pl_solver_p35pf.h) -> the 29 x 35 elimination matrix, unscaled, into a workspace in HBM
//   k_focal_solve      one WAVEFRONT = one iteration: row scaling, elimination with a matrix column per lane in registers,
//                      eigenvalues of the action matrix by the lanes together (pl_eigen_wave.h), one lane per root; keeps the
//                      solutions the estimator keeps (focal >= 0, focal <= max_focal_length: absolute_pose.cc:89-95)
//   k_focal_score      one wavefront = one model: compute_msac_score(Image, ...) (utils.cc:66-98) - inlier count and the sum of
//                      the inliers' squared residuals IN CORRESPONDENCE ORDER (the score decides comparisons in the loop, so
//                      it has to be the sequential sum): the lanes evaluate 64 correspondences at a time, the inliers' residuals
//                      are then added one by one in lane order (wave-uniform loop over the ballot).
//   k_focal_score_wg   the same score by one workgroup per model (producers / ordered chain): the few refined models of a local
//                      optimisation
//   k_focal_mask       get_inliers(Image, ...) (utils.cc:385-399), one thread per correspondence.
// Round 3's form (one kernel, one lane per sample, matrices in LDS) and what each step of round 4 bought: DESIGN 4, "The focal-length estimators".
#include "pl_focal.h"
#include "pl_kernels.h"
#include "pl_solver_p35pf.h"
#include "pl_eigen_wave.h"
#include "pl_eigen_packed.h"
#include "pl_nullvec_packed.h"
#include "pl_lm_chain.inc"
#include <algorithm>
#include <atomic>

namespace pl {

namespace {

// ---- the generator: two kernels over a workspace in HBM (element e of sample it at stage[e * B + it]) ------------------------
//   k_focal_setup   one lane = one sample: draw (or read) the sample, null space of the linear constraints, the 29 equations
//                   -> rows of the elimination matrix (coalesced: consecutive lanes = consecutive samples), N, f0
//   k_focal_solve   one WAVEFRONT = one sample: elimination in registers, eigenvalues and roots by the lanes together (below)
// Round 3's single kernel (one lane per sample, 8.1 KB of LDS per sample: 16 lanes per CU whatever the phase) took 1.5 ms per
// launch and occupied 63 CUs for a batch of 1001 samples; the two kernels take 0.21 + 0.33 ms.
constexpr int kStageN = kP35WorkDoubles, kStageF0 = kStageN + 60, kStageMx = kStageF0 + 1, kStageDoubles = kStageMx + kP35Rows;

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
    // the rows unscaled, their maxima beside them: the 35 divisions of a row are done by the 35 lanes of k_focal_solve that hold its
    // columns (p35_store_row's operations, one lane per column instead of one lane for 1015 divisions)
    double *const st = g.stage + it;
    p35pf_setup_t(xs, X, [&](int r, const P35Cubic &eq) {
        double mx = 0;
#pragma unroll
        for (int c = 0; c < kP35Cols; ++c)
            mx = fmax(mx, fabs(eq.c[c]));
#pragma unroll
        for (int c = 0; c < kP35Cols; ++c)
            st[(size_t)(r * kP35Cols + c) * B] = eq.c[c];
        st[(size_t)(kStageMx + r) * B] = mx;
    }, N, f0);
    for (int e = 0; e < 60; ++e)
        g.stage[(size_t)(kStageN + e) * B + it] = N[e];
    g.stage[(size_t)kStageF0 * B + it] = f0;
}
__global__ __launch_bounds__(64) void k_focal_setup(FocalGenArgs g) { focal_setup_body(g, blockIdx.x); }
__global__ __launch_bounds__(64) void k_focal_setup_g(const FocalGenArgs *__restrict__ gs) {
    const FocalGenArgs g = gs[blockIdx.y];
    focal_setup_body(g, blockIdx.x);
}

__device__ __forceinline__ double readlane_f64(double v, int l) { // (l uniform)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// k_focal_solve: one WAVEFRONT = one sample, three stages in one launch (as three kernels they cost two more dispatches per batch,
// which is what several problems in flight on the device's hardware queues pay for: ~60 us each under 16 streams).
//   elimination   lane c holds column c of the 29 x 35 matrix in registers (58 VGPRs); per pivot every lane searches its own column,
//                 the pivot's lane decides (v_readlane), the factors f_r = entry (r, col) come from the pivot's lane one v_readlane
//                 pair each, and all 35 columns are updated at once - element for element the operations of p35pf_eliminate
//                 (pl_solver_p35pf.h), so the results are the same bits (as one lane per sample, matrices in LDS: 0.43 ms)
//   eigenvalues   of the 10 x 10 action matrix by the lanes together (pl_eigen_wave.h; one lane per sample: 0.55 ms)
//   roots         four at a time, 16 lanes per root: null vector with a matrix column per lane in registers (pl_nullvec_packed.h;
//                 round 4: lane s = root s on its own 10 x 10 working copy in LDS), pose and focal length by the group's first lane;
//                 the solutions leave in the order of the roots (ballot + v_mbcnt)
constexpr int kSolveWaves = 4;
constexpr int kSolveLds = 100 + eig_wave_doubles(10) + 190; // action matrix | eigenvalue workspace | E (50), later the roots' scratch (kRootsScratch)
static_assert(kP35ActionDoubles <= 190, "E and the roots' scratch share a region");
// Round 5: the solve stage as THREE kernels over a per-sample record in the workspace (sample-major, behind the element-major rows):
//   [action matrix 100 | eigenvalues 10 | ok | number of real eigenvalues]
//   k_focal_elim    one wavefront = one sample: the elimination, the action matrix
//   k_focal_eig     one wavefront = FOUR samples, 16 lanes each: the eigenvalues (pl_eigen_packed.h).  Inside one kernel every wavefront
//                   iterated on its own matrix with <= 10 lanes at work and every scalar of the iteration computed 64 times - 41 % of the
//                   kernel's time, its vector ALUs 70 % busy; giving the four matrices of a workgroup to one of its wavefronts (three
//                   waiting at a barrier) was slower still (5.8 -> 7.0 ms per 64 k samples): the packed iteration wants many
//                   wavefronts per SIMD, which a kernel of its own has (18 KB of LDS per 16 samples)
//   k_focal_roots   one wavefront = one sample, 16 lanes per root: null vector (pl_nullvec_packed.h), pose, focal length; the estimator's filter
constexpr uint32_t kSplitSamples = 4096; // launches of at least so many samples take the three kernels, smaller ones the single kernel
constexpr int kActDoubles = 112, kActEv = 100, kActOk = 110, kActRoots = 111;
__device__ __forceinline__ double *focal_act(const FocalGenArgs &g, uint32_t it) {
    return g.stage + (size_t)kStageDoubles * g.num_iters + (size_t)it * kActDoubles;
}
// the elimination of sample `it` by its wavefront; E (50 doubles of LDS): the rows of the action matrix that are not shifts.  false: degenerate
__device__ __forceinline__ bool focal_eliminate(const double *stage, size_t B, uint32_t it, int lane, double *E) {
    bool ok = true;
    { // ---- elimination
        const int c = lane < kP35Cols ? lane : kP35Cols - 1; // (lanes 35..63 shadow the last column: no divergence, never read)
        double w[kP35Rows];
#pragma unroll
        for (int r = 0; r < kP35Rows; ++r) { // p35_store_row: the row scaled to unit maximum
            const double mx = stage[(size_t)(kStageMx + r) * B + it];
            const double raw = stage[(size_t)(r * kP35Cols + c) * B + it];
            w[r] = mx > 0 ? raw / mx : 0.0;
        }
        uint32_t used = 0; // (uniform)
        int pivot_of = 0;  // lane k: pivot row of eliminated monomial k
#pragma unroll 1
        for (int k = 0; k < 25; ++k) {
            const int col = k < 23 ? k : k + 1; // kP35Elim
            int pr = -1;
            double best = 0;
#pragma unroll
            for (int r = 0; r < kP35Rows; ++r) {
                const double v = fabs(w[r]);
                if (!((used >> r) & 1u) && v > best)
                    best = v, pr = r;
            }
            pr = __builtin_amdgcn_readlane(pr, col);
            best = readlane_f64(best, col);
            if (pr < 0 || best < 1e-13) { // degenerate sample
                ok = false;
                break;
            }
            used |= 1u << pr;
            pivot_of = lane == k ? pr : pivot_of;
            double mine = w[0]; // w[pr] of this lane's column
#pragma unroll
            for (int r = 1; r < kP35Rows; ++r)
                mine = r == pr ? w[r] : mine;
            const double inv = 1.0 / readlane_f64(mine, col);
            const double prow = mine * inv;
#pragma unroll
            for (int r = 0; r < kP35Rows; ++r) {
                const double f = readlane_f64(w[r], col); // entry (r, col) before this pivot's update
                if (r == pr)
                    w[r] = prow;
                else if (f != 0)
                    w[r] -= f * prow;
            }
        }
        if (ok) {
            // E[i][j] = entry (pivot row of monomial kP35ActionPivot[i], basis column j): lane kP35Basis[j] holds the column
            const int j = lane >= 30 ? lane - 25 : lane == 27 ? 0 : lane == 23 ? 1 : lane == 26 ? 2 : lane == 28 ? 3 : lane == 29 ? 4 : -1;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int row = __builtin_amdgcn_readlane(pivot_of, kP35ActionPivot[i]);
                double v = w[0];
#pragma unroll
                for (int r = 1; r < kP35Rows; ++r)
                    v = r == row ? w[r] : v;
                if (j >= 0 && lane < kP35Cols)
                    E[i * 10 + j] = v;
            }
        }
    }
    return ok;
}
__device__ __forceinline__ double focal_action_entry(const double *E, int e) { // p35pf_action_entry
    const int k = e / 10, j = e - 10 * k;
    const int sh = kP35Shifted[k];
    return sh >= 0 ? (j == sh ? 1.0 : 0.0) : -E[e];
}
__device__ __forceinline__ void focal_elim_body(const FocalGenArgs &g, uint32_t blk) {
    __shared__ double s_E[kSolveWaves][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    double *E = s_E[wave];
    double *act = focal_act(g, it);
    const bool ok = focal_eliminate(g.stage, g.num_iters, it, lane, E);
    if (ok) { // ---- the action matrix
        PL_WAVE_SYNC();
        for (int e = lane; e < 100; e += 64)
            act[e] = focal_action_entry(E, e);
    }
    if (lane == 0)
        act[kActOk] = ok ? 1.0 : 0.0;
}
constexpr int kEigWaves = 4, kEigLds = 144; // (eig_wave_doubles(10) = 140, padded)
static_assert(eig_wave_doubles(10) <= kEigLds, "a group's matrix and workspace");
__device__ __forceinline__ void focal_eig_body(const FocalGenArgs &g, uint32_t blk) {
    __shared__ double s_eig[kEigWaves][4][kEigLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
    const uint32_t it = (blk * kEigWaves + wave) * 4u + grp;
    const bool alive = it < g.num_iters;
    double *act = focal_act(g, alive ? it : 0u);
    const bool ok = alive && act[kActOk] != 0.0;
    double *mine = s_eig[wave][grp];
    if (ok)
        for (int e = gl; e < 100; e += 16)
            mine[e] = act[e];
    EigWave4<10> cx{mine, gl, lane};
    const int nr = pl_real_eigenvalues_packed<10>(cx, ok, 1e-8);
    if (ok && gl < nr)
        act[kActEv + gl] = cx.out(gl);
    if (alive && gl == 0)
        act[kActRoots] = ok ? (double)nr : 0.0;
}
// The roots of sample `it`, FOUR at a time: 16 lanes per root, lane j of a group holds column j of (action matrix - eigenvalue) in
// registers and the group finds the null vector together (pl_nullvec_packed.h: positions instead of swaps, one division per lane, the
// serial routine's bits); lane 0 of the group turns it into pose and focal length; the solutions the estimator keeps leave in the order of
// the roots.  am: the action matrix (LDS, 100), ev: the eigenvalues (nroots, ascending), ws: kRootsScratch doubles of LDS.  Returns the
// number of solutions (every lane).  (Rounds 3 - 4: one lane per root on a 10 x 10 working copy in LDS, ~2500 LDS round trips per root.)
constexpr int kRootsScratch = 60 + 4 * 10 + 10 * 8 + 10; // N | null vectors of a pass | solutions | valid
static_assert(kRootsScratch <= 190, "the single kernel's LDS block (kSolveLds)");
__device__ __forceinline__ uint32_t focal_emit_roots(const FocalGenArgs &g, uint32_t it, int lane, const double *am, const double *ev, int nroots,
                                                     double *ws) {
    const size_t B = g.num_iters;
    const int grp = lane >> 4, gl = lane & 15;
    double *Ns = ws, *vs = Ns + 60, *sols = vs + 40, *valid_s = sols + 80;
    if (lane < 60)
        Ns[lane] = g.stage[(size_t)(kStageN + lane) * B + it];
    const double f0 = g.stage[(size_t)kStageF0 * B + it];
    PL_WAVE_SYNC();
    for (int first = 0; first < nroots; first += 4) { // (uniform)
        const int root = first + grp;
        const bool on = root < nroots;
        const double e = ev[on ? root : 0];
        NullWave4<10> cx;
        cx.gl = gl, cx.lane = lane, cx.cp = 0, cx.yv = 0, cx.t1 = 0, cx.t2 = 0;
#pragma unroll
        for (int r = 0; r < 10; ++r) { // column gl of wk = am - ev I (p35pf_pose_of_root)
            const double a = gl < 10 ? am[r * 10 + gl] : 0.0;
            cx.c[r] = r == gl ? a - e : a;
        }
        pl_null_vector_packed<10>(cx, on);
        if (gl < 10)
            vs[grp * 10 + gl] = cx.yv;
        PL_WAVE_SYNC();
        if (gl == 0 && on) {
            P35Solution sol;
            bool valid = p35pf_pose_from_null_vector(vs + grp * 10, Ns, f0, sol);
            if (valid && !g.keep_all) { // the estimator's filter (absolute_pose.cc:89-95)
                if (sol.focal < 0)
                    valid = false;
                if (g.max_focal >= 0 && sol.focal > g.max_focal)
                    valid = false;
            }
            double *o = sols + root * 8;
            o[0] = sol.q.w, o[1] = sol.q.x, o[2] = sol.q.y, o[3] = sol.q.z;
            o[4] = sol.t.x, o[5] = sol.t.y, o[6] = sol.t.z, o[7] = sol.focal;
            valid_s[root] = valid ? 1.0 : 0.0;
        }
        PL_WAVE_SYNC();
    }
    const bool valid = lane < nroots && valid_s[lane < nroots ? lane : 0] != 0.0;
    const uint64_t mask = __builtin_amdgcn_ballot_w64(valid);
    if (valid) {
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        FocalModel o;
        const double *sv = sols + lane * 8;
        o.q[0] = sv[0], o.q[1] = sv[1], o.q[2] = sv[2], o.q[3] = sv[3];
        o.t[0] = sv[4], o.t[1] = sv[5], o.t[2] = sv[6];
        o.f = sv[7];
        g.models[(size_t)it * kFocalMaxModels + pos] = o;
        if (g.host_models)
            g.host_models[(size_t)it * kFocalMaxModels + pos] = o;
    }
    return (uint32_t)__popcll(mask);
}
// The solve stage in ONE kernel (rounds 4 - 5: small launches - a single problem's batch of ~10^3 samples does not fill the device, and
// what counts is the length of the chain: one wavefront per sample through all three stages, its eigenvalues by itself (pl_eigen_wave.h),
// is 0.44 ms; the three kernels below are 0.53 ms for such a launch and 20 % faster for the 64 k samples of a group)
__device__ __forceinline__ void focal_solve_body(const FocalGenArgs &g, uint32_t blk) {
    __shared__ double s_solve[kSolveWaves][kSolveLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    double *base = s_solve[wave], *eig = base + 100, *E = eig + eig_wave_doubles(10);
    uint32_t m = 0;
    if (focal_eliminate(g.stage, g.num_iters, it, lane, E)) {
        // ---- action matrix (kept for the roots) and a copy for the eigenvalue iteration, which destroys it
        PL_WAVE_SYNC();
        for (int e = lane; e < 100; e += 64) {
            const double v = focal_action_entry(E, e);
            base[e] = v;
            eig[e] = v;
        }
        const int nroots = pl_real_eigenvalues_wave<10>(eig, 1e-8, lane);
        PL_WAVE_SYNC();
        // (the eigenvalues stand at eig[130 ...], the roots' scratch behind the eigenvalue workspace)
        m = focal_emit_roots(g, it, lane, base, eig + 100 + 30, nroots, eig + eig_wave_doubles(10));
    }
    if (lane == 0) {
        g.num_models[it] = m;
        if (g.host_num_models)
            g.host_num_models[it] = m;
    }
}
// k_focal_roots: one wavefront = one sample - the action matrix and the eigenvalues from the sample's record, then focal_emit_roots
constexpr int kRootsLds = 100 + 10 + kRootsScratch; // action matrix | eigenvalues | scratch
__device__ __forceinline__ void focal_roots_body(const FocalGenArgs &g, uint32_t blk) {
    __shared__ double s_roots[kSolveWaves][kRootsLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    const double *act = focal_act(g, it);
    double *am = s_roots[wave], *ev = am + 100;
    uint32_t m = 0;
    const int nroots = (int)act[kActRoots]; // (0: degenerate sample, or no real eigenvalue)
    if (nroots > 0) {
        for (int e = lane; e < 110; e += 64) // (the record: action matrix 100 | eigenvalues 10)
            am[e] = act[e];
        m = focal_emit_roots(g, it, lane, am, ev, nroots, ev + 10);
    }
    if (lane == 0) {
        g.num_models[it] = m;
        if (g.host_num_models)
            g.host_num_models[it] = m;
    }
}
#define PL_SOLVE_ATTR __launch_bounds__(64 * kSolveWaves) __attribute__((amdgpu_waves_per_eu(4, 8)))
// (the single kernel serves launches that do not fill the device: no register cap - the packed null vectors want ~150)
__global__ __launch_bounds__(64 * kSolveWaves) void k_focal_solve(FocalGenArgs g) { focal_solve_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kSolveWaves) void k_focal_solve_g(const FocalGenArgs *__restrict__ gs) {
    const FocalGenArgs g = gs[blockIdx.y];
    focal_solve_body(g, blockIdx.x);
}
__global__ PL_SOLVE_ATTR void k_focal_elim(FocalGenArgs g) { focal_elim_body(g, blockIdx.x); }
__global__ PL_SOLVE_ATTR void k_focal_elim_g(const FocalGenArgs *__restrict__ gs) {
    const FocalGenArgs g = gs[blockIdx.y];
    focal_elim_body(g, blockIdx.x);
}
__global__ __launch_bounds__(64 * kEigWaves) void k_focal_eig(FocalGenArgs g) { focal_eig_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kEigWaves) void k_focal_eig_g(const FocalGenArgs *__restrict__ gs) {
    const FocalGenArgs g = gs[blockIdx.y];
    focal_eig_body(g, blockIdx.x);
}
__global__ __launch_bounds__(64 * kSolveWaves) void k_focal_roots(FocalGenArgs g) { focal_roots_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kSolveWaves) void k_focal_roots_g(const FocalGenArgs *__restrict__ gs) {
    const FocalGenArgs g = gs[blockIdx.y];
    focal_roots_body(g, blockIdx.x);
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

size_t focal_stage_bytes(uint32_t num_iters) { return sizeof(double) * (size_t)(kStageDoubles + kActDoubles) * num_iters; }

hipError_t launch_focal_generate(const FocalGenArgs &g, hipStream_t stream) {
    if (g.num_iters == 0)
        return hipSuccess;
    if (!g.stage)
        return hipErrorInvalidValue;
    k_focal_setup<<<dim3((g.num_iters + 63u) / 64u), dim3(64), 0, stream>>>(g);
    if (g.num_iters < kSplitSamples) {
        k_focal_solve<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
        return hipGetLastError();
    }
    k_focal_elim<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
    k_focal_eig<<<dim3((g.num_iters + 4 * kEigWaves - 1) / (4 * kEigWaves)), dim3(64 * kEigWaves), 0, stream>>>(g);
    k_focal_roots<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
    return hipGetLastError();
}
// ---- group launches (driver_focal_group.inc): blockIdx.y = member, the grid's x extent = the largest member's; `args` is a
// device-resident table of G entries.  Same bodies as the single-problem kernels: a member's results do not depend on its group.
hipError_t launch_focal_generate_g(const FocalGenArgs *args, uint32_t G, uint32_t max_iters, hipStream_t stream) {
    if (G == 0 || max_iters == 0)
        return hipSuccess;
    k_focal_setup_g<<<dim3((max_iters + 63u) / 64u, G), dim3(64), 0, stream>>>(args);
    if ((size_t)max_iters * G < kSplitSamples) { // (the same bits either way: tests/test_zz_gpu_focal_group.py)
        k_focal_solve_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
        return hipGetLastError();
    }
    k_focal_elim_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
    k_focal_eig_g<<<dim3((max_iters + 4 * kEigWaves - 1) / (4 * kEigWaves), G), dim3(64 * kEigWaves), 0, stream>>>(args);
    k_focal_roots_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
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
