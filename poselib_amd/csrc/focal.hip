// poselib_amd - kernels of the focal-length estimator (ransac_pnpf: robust/ransac.cc:58-75, FocalAbsolutePoseEstimator).
//
//   k_focal_generate   one lane = one RANSAC iteration: draw the sample of four correspondences from the iteration's position in
//                      the splitmix64 stream, solve P3.5Pf (pl_solver_p35pf.h), keep the solutions the estimator keeps
//                      (focal >= 0, focal <= max_focal_length: absolute_pose.cc:89-95).  The 29 x 35 elimination matrices live in
//                      LDS, element-major, 16 samples per workgroup (one workgroup per CU) - in HBM (first version: 5.5 ms per
//                      batch of 4096 samples) every step of the elimination paid the memory latency.
//   k_focal_score      one wavefront = one model: compute_msac_score(Image, ...) (utils.cc:66-98) - inlier count and the sum of
//                      the inliers' squared residuals IN CORRESPONDENCE ORDER (the score decides comparisons in the loop, so
//                      it has to be the sequential sum): the lanes evaluate 64 correspondences at a time, the inliers' residuals
//                      are then added one by one in lane order (wave-uniform loop over the ballot).
//   k_focal_mask       get_inliers(Image, ...) (utils.cc:385-399), one thread per correspondence.
// A first, correct device path for this estimator: neither kernel is tuned (no pre-filter, one lane per sample), DESIGN §4 has the measured numbers and what bounds them.
#include "pl_focal.h"
#include "pl_kernels.h"
#include "pl_solver_p35pf.h"
#include <atomic>

namespace pl {

namespace {

// kGenLanes samples per workgroup: their elimination matrices (8.1 KB each) fill the CU's LDS
constexpr int kGenLanes = 16;
__global__ __launch_bounds__(64) void k_focal_generate(FocalGenArgs g) {
    extern __shared__ double s_work[]; // kP35WorkDoubles x kGenLanes, element-major
    const uint32_t it = blockIdx.x * kGenLanes + threadIdx.x;
    if (it >= g.num_iters)
        return;
    uint32_t idx[kFocalSample];
    if (g.samples) { // PROSAC: drawn on the host
        for (int k = 0; k < kFocalSample; ++k)
            idx[k] = g.samples[(size_t)it * kFocalSample + k];
    } else {
        draw_sample<kFocalSample>(g.seed, g.pos_base + g.positions[it], g.n, idx);
    }
    double xs[8];
    Vec3 X[4];
    for (int k = 0; k < 4; ++k) {
        xs[2 * k] = g.a[0][idx[k]];
        xs[2 * k + 1] = g.a[1][idx[k]];
        X[k] = v3(g.a[2][idx[k]], g.a[3][idx[k]], g.a[4][idx[k]]);
    }
    P35Solution sol[kFocalMaxModels];
    const int n = p35pf(xs, X, P35Work{s_work + threadIdx.x, (size_t)kGenLanes}, sol);
    uint32_t m = 0;
    for (int i = 0; i < n; ++i) {
        if (sol[i].focal < 0)
            continue;
        if (g.max_focal >= 0 && sol[i].focal > g.max_focal)
            continue;
        FocalModel &o = g.models[(size_t)it * kFocalMaxModels + m];
        o.q[0] = sol[i].q.w, o.q[1] = sol[i].q.x, o.q[2] = sol[i].q.y, o.q[3] = sol[i].q.z;
        o.t[0] = sol[i].t.x, o.t[1] = sol[i].t.y, o.t[2] = sol[i].t.z;
        o.f = sol[i].focal;
        ++m;
    }
    g.num_models[it] = m;
}

// minimal problems given explicitly (pl_p35pf, pl_solve_focal_batch): in = count x [x 4 x 2 | X 4 x 3]; every solution is kept
__global__ __launch_bounds__(64) void k_focal_solve(const double *in, uint32_t count, FocalModel *models, uint32_t *num_models) {
    extern __shared__ double s_work[];
    const uint32_t it = blockIdx.x * kGenLanes + threadIdx.x;
    if (it >= count)
        return;
    const double *p = in + (size_t)it * 20;
    double xs[8];
    Vec3 X[4];
    for (int k = 0; k < 8; ++k)
        xs[k] = p[k];
    for (int k = 0; k < 4; ++k)
        X[k] = v3(p[8 + 3 * k], p[9 + 3 * k], p[10 + 3 * k]);
    P35Solution sol[kFocalMaxModels];
    const int n = p35pf(xs, X, P35Work{s_work + threadIdx.x, (size_t)kGenLanes}, sol);
    for (int i = 0; i < n; ++i) {
        FocalModel &o = models[(size_t)it * kFocalMaxModels + i];
        o.q[0] = sol[i].q.w, o.q[1] = sol[i].q.x, o.q[2] = sol[i].q.y, o.q[3] = sol[i].q.z;
        o.t[0] = sol[i].t.x, o.t[1] = sol[i].t.y, o.t[2] = sol[i].t.z;
        o.f = sol[i].focal;
    }
    num_models[it] = (uint32_t)n;
}

constexpr int kFocalScoreThreads = 256;

__global__ __launch_bounds__(kFocalScoreThreads) void k_focal_score(FocalScoreArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * (kFocalScoreThreads / 64) + (threadIdx.x >> 6);
    if (slot >= a.num_slots)
        return;
    if (a.num_models && (slot % kFocalMaxModels) >= a.num_models[slot / kFocalMaxModels])
        return; // (wave-uniform)
    const FocalModel m = a.models[slot];
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

__global__ void k_focal_mask(const double *x, const double *y, const double *X, const double *Y, const double *Z, uint32_t n,
                             FocalModel m, double thr2, uint8_t *mask, uint8_t *host_mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double R[9];
    focal_rotation(m, R);
    const uint8_t v = focal_reproj_mask(R, m.t, m.f, x[i], y[i], X[i], Y[i], Z[i], thr2) ? 1 : 0;
    mask[i] = v;
    if (host_mask)
        host_mask[i] = v;
}

} // namespace

hipError_t launch_focal_generate(const FocalGenArgs &g, hipStream_t stream) {
    if (g.num_iters == 0)
        return hipSuccess;
    constexpr size_t bytes = sizeof(double) * kP35WorkDoubles * kGenLanes; // 129.9 KB of the CU's 160 KB
    static std::atomic<int> prepared_dev[64]; // per device ordinal: the attribute is per-device state on some runtimes
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    std::atomic<int> &prepared = prepared_dev[dev_ & 63];
    if (!prepared.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_focal_generate),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess)
            return e;
        prepared.store(1, std::memory_order_release);
    }
    k_focal_generate<<<dim3((g.num_iters + kGenLanes - 1) / kGenLanes), dim3(kGenLanes), bytes, stream>>>(g);
    return hipGetLastError();
}
hipError_t launch_focal_solve(const double *in, uint32_t count, FocalModel *models, uint32_t *num_models, hipStream_t stream) {
    if (count == 0)
        return hipSuccess;
    constexpr size_t bytes = sizeof(double) * kP35WorkDoubles * kGenLanes;
    static std::atomic<int> prepared_dev[64]; // per device ordinal: the attribute is per-device state on some runtimes
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    std::atomic<int> &prepared = prepared_dev[dev_ & 63];
    if (!prepared.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_focal_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes);
        if (e != hipSuccess)
            return e;
        prepared.store(1, std::memory_order_release);
    }
    k_focal_solve<<<dim3((count + kGenLanes - 1) / kGenLanes), dim3(kGenLanes), bytes, stream>>>(in, count, models, num_models);
    return hipGetLastError();
}
hipError_t launch_focal_score(const FocalScoreArgs &a, hipStream_t stream) {
    if (a.num_slots == 0)
        return hipSuccess;
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
