// poselib_amd — the 5-point relative-pose generator for gfx950 (MI355X, wave64): RelativePoseEstimator::generate_models
// (estimators/relative_pose.cc:48-56) = relpose_5pt (solvers/relpose_5pt.cc:159-409) for every iteration of a batch.
// Build like kernels.hip: -O3 -ffp-contract=off (every operation rounds like the reference's SSE2 build).
//
// Three stages over a structure-of-arrays workspace ([field][iteration], coalesced, one lane per iteration), so that no
// kernel carries the live state of another - the 10 x 20 elimination of the front end, the Sturm chain of the root finder
// and the pose recovery each get the register file to themselves:
//
//   k_rel_front   straight-line code: sample -> unit bearings -> 9 x 5 epipolar matrix -> null space (full-pivot
//                 Householder, register resident: pivot rows are exchanged by comparison-selected swaps, never through a
//                 run-time index) -> 10 x 20 constraints -> 10 x 10 LU -> the 3 x 3 polynomial matrix Az.  Hands over
//                 bearings[30], nb[36], Az[39].
//   k_rel_roots   degree-10 determinant -> Sturm chain -> isolation by bisection (deferred halves in LDS, one column per
//                 lane; leaves in the workspace) -> Ridders + Newton on every leaf (sturm.h:153-231) -> back substitution and
//                 the essential matrix of every root (relpose_5pt.cc:355-392; in two passes so that Az and nb are never
//                 live together).  Hands over E[root][9].
//   k_rel_poses   the 256 iterations of a workgroup bucketed by root count in LDS (the lanes of a wavefront loop equally
//                 long): E -> four motion candidates -> cheirality on the five sample points (essential.cc:103-169).  The
//                 few poses that survive (0.57 per iteration) are only QUEUED in LDS by the root loop; the records - R(q),
//                 E = [t]x R, fp32 shadow, 24 stores - are then written by one lane per surviving pose instead of by every
//                 root iteration of every wavefront that has one survivor somewhere.  Round 3: 77 -> 40 us per 100 k
//                 iterations on a full device (E and the bearings arrive ready, no record path in the loop).
//
// Measured on MI355X with the device FILLED (16 problems x 100 k iterations per launch - what a group launch is; a single
// 100 k launch is 1563 wavefronts for 2560 slots and only shows the slowest wavefront): round 2's three kernels 244 us per
// 100 k iterations, this file 231 (front 81, roots 115 incl. the essential matrices, poses 40).  Measured and NOT adopted
// (scripts/exp/genbench.cc, bit-identical checksums throughout):
//   * root isolation as a work list shared by the 64 slots of a wavefront (ring in LDS, chain coefficients gathered from the
//     workspace): -35 % vector instructions (a lane's walk costs every wavefront its longest lane: 33 midpoint evaluations
//     per wavefront round against a mean of 11.8), but every round then waits for 30 gathers and the stage handovers of
//     1.6 M iterations (3 GB) go through HBM: 2.3x SLOWER on a full device;
//   * iterations sorted by their number of real roots before the root finder: the bisection depth hardly depends on it
//     (mean wave maximum 30.9 instead of 33.4 evaluations) and slot-sorted lanes read their iteration's fields
//     uncoalesced (4x the traffic);
//   * the essential matrices in a kernel of their own (HBM bound: 24 us), or computed inside the pose loop (346
//     registers, one wavefront per SIMD: no gain over round 2);
//   * pivot-row exchanges of the 10 x 10 LU as branches instead of selects (v_mov_b64 / v_accvgpr_mov triples and 312
//     spilled registers instead of 3600 v_cndmask: same instruction count).
// Which lane works on which iteration never influences a result: everything is addressed by iteration.
#include "pl_device.h"
#include "pl_solver_rel.h"

#include <cstdlib>

namespace pl {

// ---- workspace ------------------------------------------------------------------------------------------------------
constexpr int kRelNb = 36, kRelAz = 39, kRelBear = 30, kRelRoots = 10, kRelLeaves = 2 * kSturmSlots, kRelEss = 90;
constexpr int kRelDoubles = kRelNb + kRelAz + kRelBear + kRelRoots + kRelLeaves + kRelEss;
__host__ __device__ inline uint32_t rel_pitch(uint32_t num_iters) { return (num_iters + 255u) / 256u * 256u; }
size_t rel_stage_bytes(uint32_t num_iters) {
    const size_t P = rel_pitch(num_iters);
    return sizeof(double) * kRelDoubles * P + sizeof(uint32_t) * P;
}
struct RelStage { // everything [field][iteration]
    double *nb, *az, *bear, *roots, *leaves;
    double *ess;      // [root][9] essential matrices (row-major) of the real roots
    uint32_t *nroots; // real roots found
    uint32_t P;
};
__host__ __device__ inline RelStage rel_stage(void *stage, uint32_t num_iters) {
    RelStage s;
    s.P = rel_pitch(num_iters);
    double *d = static_cast<double *>(stage);
    s.nb = d, d += (size_t)kRelNb * s.P;
    s.az = d, d += (size_t)kRelAz * s.P;
    s.bear = d, d += (size_t)kRelBear * s.P;
    s.roots = d, d += (size_t)kRelRoots * s.P;
    s.leaves = d, d += (size_t)kRelLeaves * s.P;
    s.ess = d, d += (size_t)kRelEss * s.P;
    s.nroots = reinterpret_cast<uint32_t *>(d);
    return s;
}

// ---- stage 1 --------------------------------------------------------------------------------------------------------
#ifndef PL_FRONT_THREADS
#define PL_FRONT_THREADS 64
#endif
constexpr int kFrontThreads = PL_FRONT_THREADS;
__device__ __forceinline__ void rel_front_body(const GenerateArgs &g, const RelStage &w) {
    const uint32_t it = blockIdx.x * kFrontThreads + threadIdx.x;
    if (it >= g.num_iters)
        return;
    uint32_t idx[5];
    sample_of_iteration<5>(g, it, idx);
    Vec3 b1[5], b2[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        b1[k] = bearing(g.pts.a[0][idx[k]], g.pts.a[1][idx[k]]);
        b2[k] = bearing(g.pts.a[2][idx[k]], g.pts.a[3][idx[k]]);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double *o = w.bear + (size_t)(6 * k) * w.P + it;
        o[0] = b1[k].x, o[(size_t)w.P] = b1[k].y, o[(size_t)2 * w.P] = b1[k].z;
        o[(size_t)3 * w.P] = b2[k].x, o[(size_t)4 * w.P] = b2[k].y, o[(size_t)5 * w.P] = b2[k].z;
    }
    double nb[36], Az[3][13];
    rel5_front(b1, b2, nb, Az);
#ifdef PL_FRONT_NOSTORE // experiment builds: the arithmetic without its 75 result stores (box-state hunt, profiles/r05_box_state.md)
    {
        double chk = 0.0; // (depends on every result: nothing of the arithmetic is dead)
#pragma unroll
        for (int e = 0; e < 36; ++e)
            chk += nb[e];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 13; ++k)
                chk += Az[i][k];
        if (chk != 12345.678)
            return;
    }
#endif
#pragma unroll
    for (int e = 0; e < 36; ++e)
        w.nb[(size_t)e * w.P + it] = nb[e];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 13; ++k)
            w.az[(size_t)(i * 13 + k) * w.P + it] = Az[i][k];
}
__global__ __launch_bounds__(kFrontThreads) void k_rel_front(GenerateArgs g) { rel_front_body(g, rel_stage(g.stage, g.num_iters)); }
__global__ __launch_bounds__(kFrontThreads) void k_rel_front_g(const GroupArgs *ga) {
    const GroupArgs &gg = ga[blockIdx.z];
    if (!gg.active || blockIdx.x * (uint32_t)kFrontThreads >= gg.gen.num_iters)
        return;
    rel_front_body(gg.gen, rel_stage(gg.gen.stage, gg.gen.num_iters));
}

// ---- stage 2 --------------------------------------------------------------------------------------------------------
struct SturmWorkDev { // deferred halves: one column per lane of [slot][64] LDS arrays; leaves: the workspace
    static constexpr int kStackCap = kSturmSlots;
    double *sa, *sb;  // LDS
    unsigned *si;     // LDS
    double *leaves;   // global, [2 * slot + {0, 1}][P], this iteration's column
    size_t P;
    __device__ void push(int i, double a, double b, unsigned info) { sa[i * 64] = a, sb[i * 64] = b, si[i * 64] = info; }
    __device__ void pop(int i, double &a, double &b, unsigned &info) const { a = sa[i * 64], b = sb[i * 64], info = si[i * 64]; }
    __device__ void leaf_set(int i, double a, double b) { leaves[(size_t)(2 * i) * P] = a, leaves[(size_t)(2 * i + 1) * P] = b; }
    __device__ void leaf_get(int i, double &a, double &b) const { a = leaves[(size_t)(2 * i) * P], b = leaves[(size_t)(2 * i + 1) * P]; }
};
__device__ __forceinline__ void rel_roots_body_v1(uint32_t num_iters, const RelStage &w) {
    __shared__ double s_stack_a[kSturmSlots][64], s_stack_b[kSturmSlots][64];
    __shared__ unsigned s_stack_i[kSturmSlots][64];
    const uint32_t it = blockIdx.x * 64 + threadIdx.x;
    if (it >= num_iters)
        return;
    double Az[3][13];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 13; ++k)
            Az[i][k] = w.az[(size_t)(i * 13 + k) * w.P + it];
    double c[11];
    rel5_poly(Az, c);
    double roots[10];
    SturmWorkDev work{&s_stack_a[0][threadIdx.x], &s_stack_b[0][threadIdx.x], &s_stack_i[0][threadIdx.x], w.leaves + it, w.P};
#if defined(PL_ROOTS_STOP) && PL_ROOTS_STOP == 3 // experiment builds (scripts/exp): time the stages of the kernel
    {
        Sturm10 S_;
        double bound_;
        int sa_, sb_;
        w.nroots[it] = (uint32_t)sturm_prepare(c, S_, bound_, sa_, sb_);
        return;
    }
    const int n = 0;
#elif defined(PL_ROOTS_STOP) && PL_ROOTS_STOP == 4
    w.nroots[it] = (uint32_t)(c[0] + c[5] + c[10] > 0);
    return;
    const int n = 0;
#elif defined(PL_ROOTS_STOP) && PL_ROOTS_STOP == 1
    unsigned tiny_;
    const int n = sturm_isolate(c, work, tiny_);
    w.nroots[it] = (uint32_t)n;
    return;
#else
    const int n = sturm_roots_deg10(c, roots, work);
    w.nroots[it] = (uint32_t)n;
#if defined(PL_ROOTS_STOP) && PL_ROOTS_STOP == 2
    for (int r = 0; r < n; ++r)
        w.roots[(size_t)r * w.P + it] = roots[r];
    return;
#endif
    if (n == 0)
        return;
#endif
    // back substitution and the essential matrix of every root (relpose_5pt.cc:355-392).  Two passes, so that the
    // polynomial matrix (39 doubles) and the null-space basis (36) are never live together: the kernel keeps the
    // register budget of the root finder.
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 13; ++k)
            Az[i][k] = w.az[(size_t)(i * 13 + k) * w.P + it];
    double xs[10], ys[10];
#pragma unroll
    for (int r = 0; r < 10; ++r)
        if (r < n)
            rel5_xy_at_root(Az, roots[r], xs[r], ys[r]);
    double nb[36];
#pragma unroll
    for (int e = 0; e < 36; ++e)
        nb[e] = w.nb[(size_t)e * w.P + it];
#pragma unroll
    for (int r = 0; r < 10; ++r)
        if (r < n) {
            Mat3 E;
            rel5_essential_from_xyz(nb, xs[r], ys[r], roots[r], E);
            double *oe = w.ess + (size_t)(9 * r) * w.P + it;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                oe[(size_t)k * w.P] = E.m[k];
        }
}
__global__ __launch_bounds__(64) void k_rel_roots_v1(uint32_t num_iters, void *stage) { rel_roots_body_v1(num_iters, rel_stage(stage, num_iters)); }
__global__ __launch_bounds__(64) void k_rel_roots_v1_g(const GroupArgs *ga) {
    const GroupArgs &gg = ga[blockIdx.z];
    if (!gg.active || blockIdx.x * 64u >= gg.gen.num_iters)
        return;
    rel_roots_body_v1(gg.gen.num_iters, rel_stage(gg.gen.stage, gg.gen.num_iters));
}

// ---- stage 2, round 5 ---------------------------------------------------------------------------------------------------
// Where the time of the round-4 kernel (k_rel_roots_v1, kept for A/B: POSELIB_AMD_REL_ROOTS_V1=1) went, measured on a
// full device with builds that return after a stage (1.6 M iterations, profiles/r05_generator_roots_stages.md):
// determinant polynomial + Sturm chain 147 us, ISOLATION 730 us, Ridders + Newton 635 us, back substitution + essential
// matrices 271 us = 1783 us.  The isolation loop is not issue bound: 122 vector and 83 scalar instructions per round, 13
// branches, an LDS round trip per pop - ~2000 SIMD cycles per round at two wavefronts per SIMD - and the recursion-shaped
// loop spends a round on every VISIT (bisections, leaves, empty halves).  sturm_isolate_flat (pl_solver_rel.h) spends a
// round only on a Sturm evaluation and is straight-line inside; the leaves come out of order and are ranked afterwards.
// Also measured this round, and dropped (same checksums):
//   * 256 iterations per workgroup, the LEAVES counting-sorted by bracket width and dealt to the lanes (the Ridders loop
//     stops on the bracket's width: 7.2 steps on average, up to 30; host model 10.8 k -> 5.8 k instructions per 64
//     iterations): the polish stage went 635 -> 480 us, but four wavefronts that wait for each other at barriers made
//     the isolation stage 27 % slower (877 -> 1114 us): 1826 us in all;
//   * in addition the ROOTS dealt to the lanes for the back substitution: a lane then gathers Az and the null-space basis
//     of its root's iteration (75 loads over ~16 cache lines each): bound by the texture path, generator 3.56 -> 3.81 ms;
//   * -amdgpu-sched-strategy=max-ilp for this file: 3.616 against 3.610 ms.
struct SturmWorkFlat { // intervals to bisect and the leaves as found: one column per lane of LDS arrays; ordered leaves: workspace
    static constexpr int kPendCap = 6, kLeafCap = kSturmSlots;
    double *pa, *pb, *ua, *ub; // LDS
    unsigned *pi;              // LDS
    double *leaves;            // global, [2 * slot + {0, 1}][P], this iteration's column
    size_t P;
    __device__ void pend_push(int i, double a, double b, unsigned info) { pa[i * 64] = a, pb[i * 64] = b, pi[i * 64] = info; }
    __device__ void pend_pop(int i, double &a, double &b, unsigned &info) const { a = pa[i * 64], b = pb[i * 64], info = pi[i * 64]; }
    __device__ void uleaf_set(int i, double a, double b) { ua[i * 64] = a, ub[i * 64] = b; }
    __device__ void uleaf_get(int i, double &a, double &b) const { a = ua[i * 64], b = ub[i * 64]; }
    __device__ void leaf_set(int i, double a, double b) { leaves[(size_t)(2 * i) * P] = a, leaves[(size_t)(2 * i + 1) * P] = b; }
    __device__ void leaf_get(int i, double &a, double &b) const { a = leaves[(size_t)(2 * i) * P], b = leaves[(size_t)(2 * i + 1) * P]; }
};
__device__ __forceinline__ void rel_roots_body(uint32_t num_iters, const RelStage &w) {
    __shared__ double s_pend_a[SturmWorkFlat::kPendCap][64], s_pend_b[SturmWorkFlat::kPendCap][64];
    __shared__ unsigned s_pend_i[SturmWorkFlat::kPendCap][64];
    __shared__ double s_leaf_a[SturmWorkFlat::kLeafCap][64], s_leaf_b[SturmWorkFlat::kLeafCap][64];
    const uint32_t it = blockIdx.x * 64 + threadIdx.x;
    if (it >= num_iters)
        return;
    double Az[3][13];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 13; ++k)
            Az[i][k] = w.az[(size_t)(i * 13 + k) * w.P + it];
    double c[11];
    rel5_poly(Az, c);
    double roots[10];
    SturmWorkFlat work{&s_pend_a[0][threadIdx.x], &s_pend_b[0][threadIdx.x], &s_leaf_a[0][threadIdx.x], &s_leaf_b[0][threadIdx.x],
                       &s_pend_i[0][threadIdx.x], w.leaves + it, w.P};
    const int n = sturm_roots_deg10_flat(c, roots, work);
    w.nroots[it] = (uint32_t)n;
    if (n == 0)
        return;
    // back substitution and the essential matrix of every root (relpose_5pt.cc:355-392).  Two passes, so that the
    // polynomial matrix (39 doubles) and the null-space basis (36) are never live together.
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 13; ++k)
            Az[i][k] = w.az[(size_t)(i * 13 + k) * w.P + it];
    double xs[10], ys[10];
#pragma unroll
    for (int r = 0; r < 10; ++r)
        if (r < n)
            rel5_xy_at_root(Az, roots[r], xs[r], ys[r]);
    double nb[36];
#pragma unroll
    for (int e = 0; e < 36; ++e)
        nb[e] = w.nb[(size_t)e * w.P + it];
#pragma unroll
    for (int r = 0; r < 10; ++r)
        if (r < n) {
            Mat3 E;
            rel5_essential_from_xyz(nb, xs[r], ys[r], roots[r], E);
            double *oe = w.ess + (size_t)(9 * r) * w.P + it;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                oe[(size_t)k * w.P] = E.m[k];
        }
}
__global__ __launch_bounds__(64) void k_rel_roots(uint32_t num_iters, void *stage) { rel_roots_body(num_iters, rel_stage(stage, num_iters)); }
__global__ __launch_bounds__(64) void k_rel_roots_g(const GroupArgs *ga) {
    const GroupArgs &gg = ga[blockIdx.z];
    if (!gg.active || blockIdx.x * 64u >= gg.gen.num_iters)
        return;
    rel_roots_body(gg.gen.num_iters, rel_stage(gg.gen.stage, gg.gen.num_iters));
}

// ---- stage 3 --------------------------------------------------------------------------------------------------------
// back substitution and the essential matrix of every root (relpose_5pt.cc:355-392)

// ---- stage 4 --------------------------------------------------------------------------------------------------------
// Lane assignment: the work of an iteration is proportional to its number of real roots (0, 2, 4, ..., 10; 4.2 on
// average, but the maximum over 64 neighbours is 6.8), so the 256 iterations of a workgroup are bucketed by root count
// in LDS and every lane takes the iteration at its position of the sorted list.
constexpr int kPosesThreads = 256;
constexpr uint32_t kPoseQueue = 512; // surviving poses a workgroup can queue (more: written by the root loop itself)
__device__ __forceinline__ uint32_t rel_poses_sorted_iteration(const uint32_t *nroots_in, uint32_t num_iters) {
    __shared__ uint32_t s_cnt[12];
    __shared__ uint16_t s_perm[kPosesThreads];
    const uint32_t it0 = blockIdx.x * kPosesThreads, tid = threadIdx.x;
    const uint32_t it = it0 + tid;
    const uint32_t ne = it < num_iters ? min(nroots_in[it], 10u) : 11u; // 11: not an iteration (sorted to the end)
    if (tid < 12)
        s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t rank = atomicAdd(&s_cnt[ne], 1u);
    __syncthreads();
    uint32_t base = 0; // iterations with more roots first
    for (uint32_t k = 0; k < 12; ++k) {
        const uint32_t key = (k == 11) ? 11u : 10u - k; // order of the buckets: 10, 9, ..., 0, then the padding
        if (key == ne)
            break;
        base += s_cnt[key];
    }
    s_perm[base + rank] = (uint16_t)tid;
    __syncthreads();
    return it0 + s_perm[tid];
}

__device__ __forceinline__ void rel_poses_body(const GenerateArgs &g, const RelStage &w) {
    __shared__ double s_pose[7][kPoseQueue];
    __shared__ uint32_t s_where[kPoseQueue]; // (iteration - first iteration of the workgroup) | record slot << 8
    __shared__ uint32_t s_npose;
    __shared__ uint32_t s_nan[kPosesThreads]; // NaN records per iteration of the workgroup (statistics)
    const uint32_t it0 = blockIdx.x * kPosesThreads, tid = threadIdx.x;
    s_nan[tid] = 0;
    if (tid == 0)
        s_npose = 0;
    const uint32_t it = rel_poses_sorted_iteration(w.nroots, g.num_iters); // (ends with a barrier)
    const bool valid = it < g.num_iters;
    const int max_out = (int)g.slots_per_iter;
    int n = 0;
    if (valid) {
        const int ne = (int)w.nroots[it];
        if (ne > 0) {
            Vec3 b1[5], b2[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const double *o = w.bear + (size_t)(6 * k) * w.P + it;
                b1[k] = v3(o[0], o[(size_t)w.P], o[(size_t)2 * w.P]);
                b2[k] = v3(o[(size_t)3 * w.P], o[(size_t)4 * w.P], o[(size_t)5 * w.P]);
            }
            double *rec = g.models + (size_t)it * g.slots_per_iter * kModelStride;
#pragma unroll 1
            for (int s = 0; s < ne; ++s) { // relpose_5pt_records, root by root
                Mat3 E;
                const double *o = w.ess + (size_t)(9 * s) * w.P + it;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    E.m[k] = o[(size_t)k * w.P];
                motion_from_essential_emit<5>(E, b1, b2, [&](Quat q, Vec3 t) {
                    if (n < max_out) {
                        const uint32_t e = atomicAdd(&s_npose, 1u);
                        if (e < kPoseQueue) {
                            s_pose[0][e] = q.w, s_pose[1][e] = q.x, s_pose[2][e] = q.y, s_pose[3][e] = q.z;
                            s_pose[4][e] = t.x, s_pose[5][e] = t.y, s_pose[6][e] = t.z;
                            s_where[e] = (it - it0) | ((uint32_t)n << 8);
                        } else if (store_pose_model_q(rec + n * kModelStride, q, t, true)) {
                            atomicAdd(&s_nan[it - it0], 1u);
                        }
                    }
                    ++n;
                });
            }
        }
    }
    __syncthreads();
    // the records of the queued poses, one lane each
    const uint32_t queued = min(s_npose, kPoseQueue);
    for (uint32_t e = tid; e < queued; e += kPosesThreads) {
        const uint32_t wh = s_where[e];
        const uint32_t ite = it0 + (wh & 0xffu);
        Quat q;
        q.w = s_pose[0][e], q.x = s_pose[1][e], q.y = s_pose[2][e], q.z = s_pose[3][e];
        const Vec3 t = v3(s_pose[4][e], s_pose[5][e], s_pose[6][e]);
        double *rec = g.models + ((size_t)ite * g.slots_per_iter + (wh >> 8)) * kModelStride;
        if (store_pose_model_q(rec, q, t, true))
            atomicAdd(&s_nan[wh & 0xffu], 1u);
    }
    __syncthreads();
    uint32_t n_nan = valid ? s_nan[it - it0] : 0u;
    if (n > max_out) {
        g.ctl->gen_overflow = 1;
        n = 0;
        n_nan = 0;
    }
    if (valid)
        g.num_models[it] = (uint32_t)n;
    count_models_of_wave(g, it0, (uint32_t)n, n_nan);
}
__global__ __launch_bounds__(kPosesThreads) void k_rel_poses(GenerateArgs g) { rel_poses_body(g, rel_stage(g.stage, g.num_iters)); }
__global__ __launch_bounds__(kPosesThreads) void k_rel_poses_g(const GroupArgs *ga) {
    const GroupArgs &gg = ga[blockIdx.z];
    if (!gg.active || blockIdx.x * (uint32_t)kPosesThreads >= gg.gen.num_iters)
        return;
    rel_poses_body(gg.gen, rel_stage(gg.gen.stage, gg.gen.num_iters));
}

// ---- launchers ------------------------------------------------------------------------------------------------------
// POSELIB_AMD_REL_ROOTS_V1=1 (diagnostic, A/B): the leaf-after-leaf root kernel of rounds 2 - 4; same bits
static bool rel_roots_v1() {
    static const bool v = [] {
        const char *e = std::getenv("POSELIB_AMD_REL_ROOTS_V1");
        return e && e[0] == '1';
    }();
    return v;
}
hipError_t launch_generate_rel(const GenerateArgs &a, hipStream_t stream) {
    const uint32_t B = a.num_iters;
    k_rel_front<<<dim3((B + kFrontThreads - 1) / kFrontThreads), dim3(kFrontThreads), 0, stream>>>(a);
    if (rel_roots_v1())
        k_rel_roots_v1<<<dim3((B + 63) / 64), dim3(64), 0, stream>>>(B, a.stage);
    else
        k_rel_roots<<<dim3((B + 63) / 64), dim3(64), 0, stream>>>(B, a.stage);
    k_rel_poses<<<dim3((B + kPosesThreads - 1) / kPosesThreads), dim3(kPosesThreads), 0, stream>>>(a);
    return hipGetLastError();
}
hipError_t launch_group_generate_rel(const GroupArgs *args, uint32_t max_B, uint32_t G, hipStream_t stream) {
    k_rel_front_g<<<dim3((max_B + kFrontThreads - 1) / kFrontThreads, 1, G), dim3(kFrontThreads), 0, stream>>>(args);
    if (rel_roots_v1())
        k_rel_roots_v1_g<<<dim3((max_B + 63) / 64, 1, G), dim3(64), 0, stream>>>(args);
    else
        k_rel_roots_g<<<dim3((max_B + 63) / 64, 1, G), dim3(64), 0, stream>>>(args);
    k_rel_poses_g<<<dim3((max_B + kPosesThreads - 1) / kPosesThreads, 1, G), dim3(kPosesThreads), 0, stream>>>(args);
    return hipGetLastError();
}

} // namespace pl
