// poselib_amd - kernels of the shared-focal relative pose estimator (ransac_shared_focal_relpose: robust/ransac.cc:182-203,
// SharedFocalRelativePoseEstimator robust/estimators/relative_pose.cc:154-203, refiner robust/optim/relative.h:488-592).
//
//   k_sfocal_setup     one lane = one RANSAC iteration: the sample of six correspondences from the iteration's position in the
//                      splitmix64 stream (or the host's PROSAC sample), unit bearings, null space and the ten equations of the
//                      6-point solver (pl_solver_6ptf.h) into a workspace in HBM
//   k_sfocal_solve     one WAVEFRONT = one iteration: row reduction to the 15 x 15 companion matrix, its eigenvalues (pl_eigen_wave.h),
//                      one lane per root, then one lane per solution
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
// Round 3's form (one lane per sample throughout) and what each step of round 4 bought: DESIGN 4, "The focal-length estimators".
#include "pl_kernels.h"
#include "pl_device.h"
#include "pl_sfocal.h"
#include "pl_solver_6ptf.h"
#include "pl_eigen_wave.h"
#include "pl_eigen_packed.h"
#include "pl_nullvec_packed.h"
#include "pl_lm_chain.inc"
#include <algorithm>
#include <atomic>

namespace pl {

namespace {

// ---- the generator: two kernels over a workspace in HBM (element e of sample it at stage[e * B + it]) ------------------------
//   k_sfocal_setup   one lane = one sample: null space of the epipolar constraints and the ten equations C (written straight into the
//                    stage, coalesced), the bearings
//   k_sfocal_solve   one WAVEFRONT = one sample: row reduction of the equations to the 15 x 15 companion matrix, its eigenvalues
//                    and the roots by the lanes together (below)
// Round 3's single kernel held 8.6 KB of LDS per sample through all stages (2.86 ms on 63 CUs per batch of 1001 samples: four
// problems filled the device, which bounded the throughput of several host threads at 1.4 k problems/s).
constexpr int kStC = 0, kStNb = 300, kStX = 327, kStDoubles = 363;

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
    six_nullspace_equations(x1, x2, SixWork{st + (size_t)kStC * B, B}, nb); // (the equations straight into the stage: coalesced)
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

// k_sfocal_solve: one WAVEFRONT = one sample, three stages in one launch.
//   companion    six_companion_wave (pl_eigen_wave.h): Gaussian elimination of the w^2 part with complete pivoting, the 10 x 10 system
//                with 15 right-hand sides, the companion matrix (one lane per sample, matrices in LDS: 360 k cycles, 81 % of the old
//                setup kernel)
//   eigenvalues  the 15 x 15 companion matrix in LDS, balanced and reduced by the lanes together (pl_eigen_wave.h: six_eigenvalues of
//                pl_solver_6ptf.h, the same operations on every element; as one lane per sample: 2.1 ms per batch)
//   roots        phase 1, lane s = root s: (x, y) from the null vector of C0 + w C1 + w^2 C2 (its own 10 x 10 matrix in LDS).  Lane 0
//                then builds the list of solutions ascending in y exactly as the serial routine inserts them.  Phase 2, lane s =
//                solution s: essential matrix, up to four poses; the models leave in the order of the solutions (prefix sum of the
//                counts).  As one lane per sample (root after root, solution after solution): 0.79 ms per batch.
constexpr int kSolveWaves = 4, kFinRoots = 8, kMaxRoots = 16; // (kFinRoots: sizes the single kernel's row-reduction region - rounds 3 - 4: per-root working copies)
constexpr int kFinC = 0, kFinA = 300, kFinNb = kFinA + 100 * kFinRoots, kFinX = kFinNb + 27, kFinTmp = kFinX + 36,
              kFinDoubles = kFinTmp + 7 * kMaxRoots;
static_assert(eig_wave_doubles(15) <= 100 * kFinRoots, "the eigenvalue workspace lives in the roots' region");
// Round 5: the solve stage as THREE kernels over a per-sample record in the workspace (sample-major, behind the element-major rows):
//   [companion matrix 225 (later: the solutions) | eigenvalues 15 | ok | number of real eigenvalues | number of solutions]
//   k_sfocal_comp    one wavefront = one sample: the row reduction to the companion matrix (six_companion_wave)
//   k_sfocal_eig     one wavefront = FOUR samples, 16 lanes each: balancing and eigenvalues (pl_eigen_packed.h).  Inside one kernel every
//                    wavefront iterated on its own matrix with <= 15 lanes at work and every scalar of the iteration computed 64 times:
//                    63 % of the kernel's time (profiles/r05_focal_batch.md)
//   k_sfocal_roots   one wavefront = one sample, 16 lanes per root: (x, y) from the null vector of C0 + w C1 + w^2 C2 (pl_nullvec_packed.h), the list of solutions
//   k_sfocal_poses   one wavefront = FOUR samples, one lane per solution: essential matrix, up to four poses; the models in order
constexpr uint32_t kSplitSamples = 4096; // launches of at least so many samples take the three kernels, smaller ones the single kernel
constexpr int kSfActDoubles = 244, kSfActEv = 225, kSfActOk = 240, kSfActRoots = 241, kSfActNs = 242,
              kSfActSol = 0; // (the solutions sx | sy | sw, 16 each, take the companion matrix's place once the eigenvalues are known)
__device__ __forceinline__ double *sfocal_act(const SFocalGenArgs &g, uint32_t it) {
    return g.stage + (size_t)kStDoubles * g.num_iters + (size_t)it * kSfActDoubles;
}
constexpr int kCompLds = 792; // T (225) | Cw (300) | A (100) | B (150) | factors (16)
// the equations of sample `it` (element-major rows of the workspace) row-reduced to the companion matrix by the wavefront.
// reg: T (225) | Cw (300) | A (100) | B (150) | factors (16) of LDS.  true (uniform): T stands in reg[0 .. 225), all entries finite
__device__ __forceinline__ bool sfocal_companion(const SFocalGenArgs &g, uint32_t it, int lane, double *reg, double *keep_C) {
    const size_t B = g.num_iters;
    const double *st = g.stage + it;
    for (int e = lane; e < 300; e += 64) {
        const double v = st[(size_t)(kStC + e) * B];
        if (keep_C) // (the roots' null vectors read the equations again)
            keep_C[e] = v;
        reg[225 + e] = v;
    }
    bool have_T = false;
    if (six_companion_wave(reg + 225, reg, reg + 525, reg + 625, reg + 775, lane)) {
        bool finite = true;
        for (int e = lane; e < 225; e += 64)
            finite = finite && isfinite(reg[e]);
        have_T = !__builtin_amdgcn_ballot_w64(!finite); // (a vanishing pivot: the balancing would not terminate on an infinite entry)
    }
    return have_T;
}
__device__ __forceinline__ void sfocal_comp_body(const SFocalGenArgs &g, uint32_t blk) {
    __shared__ double s_comp[kSolveWaves][kCompLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    double *reg = s_comp[wave];
    double *act = sfocal_act(g, it);
    const bool have_T = sfocal_companion(g, it, lane, reg, nullptr);
    if (have_T)
        for (int e = lane; e < 225; e += 64)
            act[e] = reg[e];
    if (lane == 0)
        act[kSfActOk] = have_T ? 1.0 : 0.0;
}
constexpr int kEigWaves = 4, kEigLds = 288; // (eig_wave_doubles(15) = 285, padded)
static_assert(eig_wave_doubles(15) <= kEigLds, "a group's matrix and workspace");
__device__ __forceinline__ void sfocal_eig_body(const SFocalGenArgs &g, uint32_t blk) {
    __shared__ double s_eig[kEigWaves][4][kEigLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
    const uint32_t it = (blk * kEigWaves + wave) * 4u + grp;
    const bool alive = it < g.num_iters;
    double *act = sfocal_act(g, alive ? it : 0u);
    const bool ok = alive && act[kSfActOk] != 0.0;
    double *mine = s_eig[wave][grp];
    if (ok)
        for (int e = gl; e < 225; e += 16)
            mine[e] = act[e];
    EigWave4<15> cx{mine, gl, lane};
    cx.sync();
    pl_balance_pow2_packed<15>(cx, ok);
    const int nr = pl_real_eigenvalues_packed<15>(cx, ok, 1e-8);
    if (ok && gl < nr)
        act[kSfActEv + gl] = cx.out(gl);
    if (alive && gl == 0)
        act[kSfActRoots] = ok ? (double)nr : 0.0;
}
// phase 1, the roots of a sample FOUR at a time: 16 lanes per root, lane j of a group forms column j of C0 + w C1 + w^2 C2 in registers
// and the group finds the null vector together (pl_nullvec_packed.h); (x, y) of the root from its entries 7, 8, 9.  Lane 0 of the
// wavefront then builds the list of solutions ascending in y as the serial routine inserts them.  C: the equations (LDS, 300), ev: the
// eigenvalues (nroots), ws: 6 x kMaxRoots doubles of LDS - the list is left in its second half (sx | sy | sw); returns its length (uniform).
// (Rounds 3 - 4: one lane per root on a 10 x 10 working copy in LDS, ~2800 LDS round trips per root.)
__device__ __forceinline__ int sfocal_root_solutions(int lane, const double *C, const double *ev, int nroots, double *ws) {
    const int grp = lane >> 4, gl = lane & 15;
    double *rx = ws, *ry = rx + kMaxRoots, *rw = ry + kMaxRoots, *sx = rw + kMaxRoots, *sy = sx + kMaxRoots, *sw = sy + kMaxRoots;
    uint32_t fmask = 0; // (uniform) bit s: root s has a solution
    for (int first = 0; first < nroots; first += 4) { // (uniform)
        const int root = first + grp;
        const double wv = ev[root < nroots ? root : 0];
        const bool on = root < nroots && !(wv < 1e-8); // six_root_xy: focal lengths beyond 1e4 are dropped
        NullWave4<10> cx;
        cx.gl = gl, cx.lane = lane, cx.cp = 0, cx.yv = 0, cx.t1 = 0, cx.t2 = 0;
#pragma unroll
        for (int r = 0; r < 10; ++r) { // column gl of A = C0 + w (C1 + w C2)
            const int e = r * 10 + (gl < 10 ? gl : 0);
            cx.c[r] = C[e] + wv * (C[100 + e] + wv * C[200 + e]);
        }
        pl_null_vector_packed<10>(cx, on);
        const double v7 = null_row_bcast<7>(cx.yv), v8 = null_row_bcast<8>(cx.yv), v9 = null_row_bcast<9>(cx.yv);
        const bool found = on && !(v9 == 0);
        if (gl == 0 && found)
            rx[root] = v7 / v9, ry[root] = v8 / v9, rw[root] = wv;
        const uint64_t b = __builtin_amdgcn_ballot_w64(gl == 0 && found);
        fmask |= ((uint32_t)(b & 1u) | (uint32_t)((b >> 16) & 1u) << 1 | (uint32_t)((b >> 32) & 1u) << 2 | (uint32_t)((b >> 48) & 1u) << 3) << first;
        PL_WAVE_SYNC();
    }
    int ns = 0;
    if (lane == 0)
        for (int s = 0; s < nroots; ++s)
            if ((fmask >> s) & 1u)
                six_insert_solution(sx, sy, sw, ns, rx[s], ry[s], rw[s]);
    ns = __builtin_amdgcn_readfirstlane(ns);
    PL_WAVE_SYNC();
    return ns;
}
// phase 2, lane gl (< 16) of a group of 16 lanes = solution gl of sample `it`: essential matrix, up to four poses; the models leave in
// the order of the solutions (prefix sum of the counts over the group).  x1 / x2 / nb: the sample's bearings and null space, cnt: 16
// doubles of scratch - LDS of the group.  One group per wavefront (the single kernel) or four (k_sfocal_poses).  Returns the number of
// models of the sample (every lane of the group).
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
// both phases by the sample's own wavefront (the single kernel)
__device__ __forceinline__ uint32_t sfocal_emit_roots(const SFocalGenArgs &g, uint32_t it, int lane, double *base, const double *ev, int nroots) {
    const int ns = sfocal_root_solutions(lane, base + kFinC, ev, nroots, base + kFinTmp);
    double *sx = base + kFinTmp + 3 * kMaxRoots, *sy = sx + kMaxRoots, *sw = sy + kMaxRoots, *cnt = sw + kMaxRoots;
    const int s = lane < kMaxRoots ? lane : 0;
    const Vec3 *x1 = reinterpret_cast<const Vec3 *>(base + kFinX);
    return sfocal_emit_poses(g, it, lane < kMaxRoots ? lane : kMaxRoots, ns, sx[s], sy[s], sw[s], x1, x1 + 6, base + kFinNb, cnt);
}
// The solve stage in ONE kernel (small launches: see focal.hip - the chain of a single problem's batch is shorter this way; the same bits)
__device__ __forceinline__ void sfocal_solve_body(const SFocalGenArgs &g, uint32_t blk) {
    __shared__ double s_fin[kSolveWaves][kFinDoubles];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    const size_t B = g.num_iters;
    const double *st = g.stage + it;
    double *base = s_fin[wave];
    double *reg = base + kFinA; // T (225) | Cw (300) | A (100) | B (150) | factors (16): dead before the roots use the region
    uint32_t m = 0;
    if (lane < 27)
        base[kFinNb + lane] = st[(size_t)(kStNb + lane) * B];
    if (lane < 36)
        base[kFinX + lane] = st[(size_t)(kStX + lane) * B];
    if (sfocal_companion(g, it, lane, reg, base + kFinC)) {
        pl_balance_pow2_wave<15>(reg, lane);
        const int nroots = pl_real_eigenvalues_wave<15>(reg, 1e-8, lane);
        PL_WAVE_SYNC();
        if (nroots > 0) // (the eigenvalues stand at reg[270 ...])
            m = sfocal_emit_roots(g, it, lane, base, reg + 225 + 45, nroots);
    }
    if (lane == 0) {
        g.num_models[it] = m;
        if (g.host_num_models)
            g.host_num_models[it] = m;
    }
}
// k_sfocal_roots: one wavefront = one sample - the equations from the workspace, the eigenvalues from the sample's record, then
// sfocal_root_solutions; the list of solutions goes into the record
constexpr int kSfRootsLds = 300 + kMaxRoots + 6 * kMaxRoots; // equations | eigenvalues | rx ry rw | sx sy sw
__device__ __forceinline__ void sfocal_roots_body(const SFocalGenArgs &g, uint32_t blk) {
    __shared__ double s_fin[kSolveWaves][kSfRootsLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t it = blk * kSolveWaves + wave; // (wave-uniform)
    if (it >= g.num_iters)
        return;
    const size_t B = g.num_iters;
    const double *st = g.stage + it;
    double *act = sfocal_act(g, it);
    double *C = s_fin[wave], *ev = C + 300, *ws = ev + kMaxRoots;
    const int nroots = (int)act[kSfActRoots]; // (0: no companion matrix, or no real eigenvalue)
    int ns = 0;
    if (nroots > 0) {
        for (int e = lane; e < 300; e += 64) // the equations
            C[e] = st[(size_t)(kStC + e) * B];
        if (lane < nroots)
            ev[lane] = act[kSfActEv + lane];
        PL_WAVE_SYNC();
        ns = sfocal_root_solutions(lane, C, ev, nroots, ws);
        const double *sx = ws + 3 * kMaxRoots, *sy = sx + kMaxRoots, *sw = sy + kMaxRoots;
        if (lane < ns) // the list of solutions: into the record (the companion matrix's place - dead since the eigenvalue kernel)
            act[kSfActSol + lane] = sx[lane], act[kSfActSol + kMaxRoots + lane] = sy[lane], act[kSfActSol + 2 * kMaxRoots + lane] = sw[lane];
    }
    if (lane == 0)
        act[kSfActNs] = (double)ns;
}
// k_sfocal_poses: one wavefront = FOUR samples, lane 16 g + s = solution s of sample g (a sample has <= 15 solutions and typically
// one to three: as one wavefront per sample the poses were 18 % of the solve stage with a handful of lanes at work)
constexpr int kPoseWaves = 4, kPoseLds = 80; // per group: null space 27 | bearings 36 | counts 16
__device__ __forceinline__ void sfocal_poses_body(const SFocalGenArgs &g, uint32_t blk) {
    __shared__ double s_pose[kPoseWaves][4][kPoseLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
    const uint32_t it = (blk * kPoseWaves + wave) * 4u + grp;
    const bool alive = it < g.num_iters;
    const size_t B = g.num_iters;
    const double *st = g.stage + (alive ? it : 0u);
    const double *act = sfocal_act(g, alive ? it : 0u);
    const int ns = alive ? (int)act[kSfActNs] : 0;
    double *mine = s_pose[wave][grp];
    if (ns > 0)
        for (int e = gl; e < 63; e += 16)
            mine[e] = e < 27 ? st[(size_t)(kStNb + e) * B] : st[(size_t)(kStX + (e - 27)) * B];
    PL_WAVE_SYNC();
    const int s = gl < ns ? gl : 0;
    const double sxv = ns > 0 ? act[kSfActSol + s] : 0.0, syv = ns > 0 ? act[kSfActSol + kMaxRoots + s] : 0.0,
                 swv = ns > 0 ? act[kSfActSol + 2 * kMaxRoots + s] : 1.0;
    const Vec3 *x1 = reinterpret_cast<const Vec3 *>(mine + 27);
    const uint32_t m = sfocal_emit_poses(g, alive ? it : 0u, gl, ns, sxv, syv, swv, x1, x1 + 6, mine, mine + 63);
    if (alive && gl == 0) {
        g.num_models[it] = m;
        if (g.host_num_models)
            g.host_num_models[it] = m;
    }
}
#define PL_SOLVE_ATTR __launch_bounds__(64 * kSolveWaves) __attribute__((amdgpu_waves_per_eu(4, 8)))
// (the single kernel serves launches that do not fill the device: no register cap - the packed null vectors want ~140)
__global__ __launch_bounds__(64 * kSolveWaves) void k_sfocal_solve(SFocalGenArgs g) { sfocal_solve_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kSolveWaves) void k_sfocal_solve_g(const SFocalGenArgs *__restrict__ gs) {
    const SFocalGenArgs g = gs[blockIdx.y];
    sfocal_solve_body(g, blockIdx.x);
}
__global__ PL_SOLVE_ATTR void k_sfocal_comp(SFocalGenArgs g) { sfocal_comp_body(g, blockIdx.x); }
__global__ PL_SOLVE_ATTR void k_sfocal_comp_g(const SFocalGenArgs *__restrict__ gs) {
    const SFocalGenArgs g = gs[blockIdx.y];
    sfocal_comp_body(g, blockIdx.x);
}
__global__ __launch_bounds__(64 * kEigWaves) void k_sfocal_eig(SFocalGenArgs g) { sfocal_eig_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kEigWaves) void k_sfocal_eig_g(const SFocalGenArgs *__restrict__ gs) {
    const SFocalGenArgs g = gs[blockIdx.y];
    sfocal_eig_body(g, blockIdx.x);
}
__global__ __launch_bounds__(64 * kPoseWaves) void k_sfocal_poses(SFocalGenArgs g) { sfocal_poses_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kPoseWaves) void k_sfocal_poses_g(const SFocalGenArgs *__restrict__ gs) {
    const SFocalGenArgs g = gs[blockIdx.y];
    sfocal_poses_body(g, blockIdx.x);
}
__global__ __launch_bounds__(64 * kSolveWaves) void k_sfocal_roots(SFocalGenArgs g) { sfocal_roots_body(g, blockIdx.x); }
__global__ __launch_bounds__(64 * kSolveWaves) void k_sfocal_roots_g(const SFocalGenArgs *__restrict__ gs) {
    const SFocalGenArgs g = gs[blockIdx.y];
    sfocal_roots_body(g, blockIdx.x);
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

size_t sfocal_stage_bytes(uint32_t num_iters) { return sizeof(double) * (size_t)(kStDoubles + kSfActDoubles) * num_iters; }

hipError_t launch_sfocal_generate(const SFocalGenArgs &g, hipStream_t stream) {
    if (g.num_iters == 0)
        return hipSuccess;
    if (!g.stage)
        return hipErrorInvalidValue;
    k_sfocal_setup<<<dim3((g.num_iters + 63u) / 64u), dim3(64), 0, stream>>>(g);
    if (g.num_iters < kSplitSamples) {
        k_sfocal_solve<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
        return hipGetLastError();
    }
    k_sfocal_comp<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
    k_sfocal_eig<<<dim3((g.num_iters + 4 * kEigWaves - 1) / (4 * kEigWaves)), dim3(64 * kEigWaves), 0, stream>>>(g);
    k_sfocal_roots<<<dim3((g.num_iters + kSolveWaves - 1) / kSolveWaves), dim3(64 * kSolveWaves), 0, stream>>>(g);
    k_sfocal_poses<<<dim3((g.num_iters + 4 * kPoseWaves - 1) / (4 * kPoseWaves)), dim3(64 * kPoseWaves), 0, stream>>>(g);
    return hipGetLastError();
}
// ---- group launches (driver_focal_group.inc): blockIdx.y = member, the grid's x extent = the largest member's; `args` is a
// device-resident table of G entries.  Same bodies as the single-problem kernels: a member's results do not depend on its group.
hipError_t launch_sfocal_generate_g(const SFocalGenArgs *args, uint32_t G, uint32_t max_iters, hipStream_t stream) {
    if (G == 0 || max_iters == 0)
        return hipSuccess;
    k_sfocal_setup_g<<<dim3((max_iters + 63u) / 64u, G), dim3(64), 0, stream>>>(args);
    if ((size_t)max_iters * G < kSplitSamples) { // (the same bits either way: tests/test_zz_gpu_focal_group.py)
        k_sfocal_solve_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
        return hipGetLastError();
    }
    k_sfocal_comp_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
    k_sfocal_eig_g<<<dim3((max_iters + 4 * kEigWaves - 1) / (4 * kEigWaves), G), dim3(64 * kEigWaves), 0, stream>>>(args);
    k_sfocal_roots_g<<<dim3((max_iters + kSolveWaves - 1) / kSolveWaves, G), dim3(64 * kSolveWaves), 0, stream>>>(args);
    k_sfocal_poses_g<<<dim3((max_iters + 4 * kPoseWaves - 1) / (4 * kPoseWaves), G), dim3(64 * kPoseWaves), 0, stream>>>(args);
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
