// poselib_amd — k_lm_cam: Levenberg-Marquardt on (pose, camera intrinsics) for the absolute-pose bundle adjustment with
// refine_focal_length / refine_principal_point / refine_extra_params (gfx950, wave64; -ffp-contract=off like kernels.hip).
// Reference: robust/bundle.cc:93-118, robust/optim/absolute.h:49-171, misc/camera_models.cc (parameter Jacobians),
// robust/optim/lm_impl.h:56-140.  The arithmetic is pl_refine_cam.h / pl_refine.h, shared with the host test build.
//
// One workgroup of 256 lanes per task, the whole LM loop on the device.  K = 6 + M columns, M = 1..8 refined parameters:
// the normal equations have up to 119 entries, too many for per-lane accumulators (k_lm's way), so the two halves of a
// Jacobian pass take turns over rounds of 256 correspondences:
//   producers  every lane evaluates one correspondence and writes its row [w, w r0, w r1, J0[14], J1[14]] to LDS; rows of
//              correspondences without contribution (masked out, behind the camera, weight zero) are dropped by a
//              ballot + prefix compaction that keeps ascending order;
//   consumers  lane e < K(K+1)/2 + K owns entry e of [JtJ lower triangle | Jtr] and adds the round's rows to it one
//              after the other.
// Every entry is therefore summed correspondence after correspondence - the reference's order
// (jacobian_accumulator.h:82-97) - for EVERY n, not only up to kLMSeqPoints as in k_lm: the refined pose and camera
// equal the oracle's to the bit: the robust cost is summed in the reference's order as well, at every n since round 4.  The consumer loop is branch-free and takes eight rows per step (their LDS reads travel together, the
// additions stay a chain in row order); the final bundle runs once per problem, on the inliers.
#include "pl_kernels.h"
#include "pl_lm_chain.inc"
#include "pl_device.h"
#include "pl_refine_cam.h"
#include <algorithm>
#include <atomic>

namespace pl {

namespace {

constexpr int kCamThreads = 256;
constexpr int kCamWaves = kCamThreads / 64;
constexpr int kProd = kCamThreads - 64; // correspondences per round: wavefronts 1 .. 3 produce, wavefront 0 adds

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

} // namespace

__global__ __launch_bounds__(kCamThreads) void k_lm_cam(LMTask *tasks) {
    extern __shared__ __attribute__((aligned(16))) double s_rows[]; // 2 buffers x kProd rows x kCamRow
    __shared__ LMTask s_task;
    __shared__ LMControl ctl;
    __shared__ double cur[kParamDoubles], trial[kParamDoubles];
    __shared__ CameraParams cam_cur, cam_trial;
    __shared__ double s_R[9];
    __shared__ double normal[kCamMaxEntries];
    __shared__ uint32_t s_wcnt[2][kCamWaves]; // rows a producer wavefront left in its third of buffer 0 / 1
    __shared__ double s_racc;
    __shared__ uint32_t s_count;
    __shared__ int s_idx[kCamMaxParams];
    __shared__ int s_M;

    LMTask &Tout = tasks[blockIdx.x];
    {
        static_assert(sizeof(LMTask) % 8 == 0, "copied as 64-bit words");
        const uint64_t *src = reinterpret_cast<const uint64_t *>(&Tout);
        uint64_t *dst = reinterpret_cast<uint64_t *>(&s_task);
        for (uint32_t w = threadIdx.x; w < sizeof(LMTask) / 8; w += kCamThreads)
            dst[w] = src[w];
        __syncthreads();
    }
    const LMTask &T = s_task;
    const PointSet pts = T.pts;
    const uint8_t *mask = T.mask;
    const double pscale = T.point_scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    if (T.gate_count && *T.gate_count <= T.gate_min) { // (uniform) the task does not run: see LMTask
        if (threadIdx.x == 0) {
            Tout.iterations = 0;
            Tout.skipped = 2u;
        }
        return;
    }
    if (threadIdx.x == 0) {
        if (T.start_record) { // absolute pose: q, t of the chosen model's record
            for (int i = 0; i < kParamDoubles; ++i)
                cur[i] = 0.0;
            for (int i = 0; i < 7; ++i)
                cur[i] = T.start_record[i];
        } else {
            for (int i = 0; i < kParamDoubles; ++i)
                cur[i] = T.params[i];
        }
        cam_cur = T.cam;
        ctl.opt = T.opt;
        ctl.loss = make_loss(T.opt.loss_type, T.opt.loss_scale);
        ctl.done = 0;
        int idx[kCamMaxParams];
        const int M = camera_refinement_idx(T.cam.model_id, T.cam_flags, idx);
        for (int m = 0; m < kCamMaxParams; ++m)
            s_idx[m] = (m < M) ? idx[m] : 0;
        s_M = M;
    }
    __syncthreads();
    const int M = s_M, K = 6 + M;
    const int NT = K * (K + 1) / 2 + K;
    // the consumer (wavefront 0): lane e owns entry e of [JtJ lower triangle | Jtr] and, beyond 64 entries (K >= 10), entry e + 64 as well
    const CamEntry entry = cam_entry_of(min(lane, NT - 1), K, s_idx), entry_hi = cam_entry_of(min(lane + 64, NT - 1), K, s_idx);
    const bool own_lo = wave == 0 && lane < NT, own_hi = wave == 0 && lane + 64 < NT;

    auto rotation_of = [&](const double *p) {
        if (threadIdx.x == 0) {
            Quat q;
            q.w = p[0], q.x = p[1], q.y = p[2], q.z = p[3];
            const Mat3 R = quat_to_rotmat(q);
            for (int i = 0; i < 9; ++i)
                s_R[i] = R.m[i];
        }
        __syncthreads();
    };

    // Both passes are a two-stage pipeline over rounds of kProd = 192 correspondences (round 4; up to then every wavefront
    // produced a round of 256, waited, and watched wavefront 0 add): wavefronts 1 .. 3 evaluate round r into buffer r & 1 while
    // wavefront 0 adds round r - 1 from the other buffer; ONE barrier per round.  Each producer wavefront compacts ITS 64
    // correspondences into its own third of the buffer (ballot + v_mbcnt: no count has to cross wavefronts before the rows are
    // written), the consumer walks the thirds in order - ascending correspondences, as before.
    const uint32_t rounds = (pts.n + (uint32_t)kProd - 1u) / (uint32_t)kProd;
    const int pw = wave - 1; // producer wavefront 0 .. 2 (wavefront 0: consumer)

    // robust cost at (p, cam) -> s_racc, s_count: the terms in the reference's order at EVERY n - every correspondence's term to
    // LDS (zeros for skipped ones: x + 0.0 = x), lane 0 adds a round's 192 terms with the inline-asm chain of k_lm_ordered
    // (pl_lm_chain.inc: 11.7 cycles per term)
    auto cost_pass = [&](const double *p, const CameraParams &camera) {
        rotation_of(p);
        const Loss loss = ctl.loss;
        const CameraParams cam = camera;
        double *const terms = s_rows; // [2][kProd]
        double tot = 0.0;
        uint32_t cnt = 0;
        for (uint32_t r = 0; r <= rounds; ++r) {
            if (wave > 0) {
                if (r < rounds) {
                    double term = 0.0;
                    bool counted = false;
                    const uint32_t i = r * (uint32_t)kProd + (uint32_t)(pw * 64 + lane);
                    if (i < pts.n && !(mask && !mask[i]))
                        counted = abs_cam_cost(p, s_R, cam, loss, pts.a[0][i] * pscale, pts.a[1][i] * pscale, pts.a[2][i], pts.a[3][i],
                                               pts.a[4][i], term);
                    terms[(r & 1u) * kProd + pw * 64 + lane] = counted ? term : 0.0;
                    cnt += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(counted)); // (every lane holds its wavefront's count)
                }
            } else if (r > 0 && threadIdx.x == 0) {
#pragma unroll 1
                for (int q = 0; q < kProd; q += 64) {
                    const uint32_t addr = (uint32_t)(uintptr_t)&terms[((r - 1u) & 1u) * kProd + q];
                    PL_LM_CHAIN64(tot, addr);
                }
            }
            __syncthreads();
        }
        if (lane == 0)
            s_wcnt[0][wave] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            s_racc = tot;
            uint32_t c = 0;
            for (int w = 1; w < kCamWaves; ++w)
                c += s_wcnt[0][w];
            s_count = c;
        }
        __syncthreads();
    };

    // normal equations at (p, cam) -> normal[0 .. NT), s_count
    auto jacobian_pass = [&](const double *p, const CameraParams &camera) {
        rotation_of(p);
        const Loss loss = ctl.loss;
        const CameraParams cam = camera;
        double acc = 0.0, acc_hi = 0.0;
        uint32_t total = 0; // (consumer)
        for (uint32_t r = 0; r <= rounds; ++r) {
            if (wave > 0) {
                if (r < rounds) {
                    const uint32_t i = r * (uint32_t)kProd + (uint32_t)(pw * 64 + lane);
                    double row[kCamRow];
                    bool kept = false;
                    if (i < pts.n && !(mask && !mask[i]))
                        kept = abs_cam_row(p, s_R, cam, loss, pts.a[0][i] * pscale, pts.a[1][i] * pscale, pts.a[2][i], pts.a[3][i],
                                           pts.a[4][i], row);
                    const uint64_t b = __builtin_amdgcn_ballot_w64(kept);
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                    if (lane == 0)
                        s_wcnt[r & 1u][wave] = (uint32_t)__popcll(b);
                    if (kept) {
                        double *dst = s_rows + ((size_t)(r & 1u) * kProd + (size_t)pw * 64 + below) * kCamRow;
#pragma unroll
                        for (int k = 0; k < kCamRow; ++k)
                            dst[k] = row[k];
                    }
                }
            } else if (r > 0) {
                const uint32_t buf = (r - 1u) & 1u;
                for (int w = 1; w < kCamWaves; ++w) { // the three thirds in order
                    const uint32_t rows = s_wcnt[buf][w];
                    const double *const r0 = s_rows + ((size_t)buf * kProd + (size_t)(w - 1) * 64) * kCamRow;
                    // eight rows per step: their LDS reads and products are independent and travel together, only the additions
                    // into the entry are a chain - in row order, as the reference adds them (measured: 320 -> 40 cycles per row)
                    auto add_rows = [&](const CamEntry &en, double &a) {
                        const double *rp = r0;
                        uint32_t q = 0;
                        for (; q + 8u <= rows; q += 8u, rp += 8 * kCamRow) {
                            double t[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                t[u] = cam_entry_term(rp + u * kCamRow, en);
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                a += t[u];
                        }
                        for (; q < rows; ++q, rp += kCamRow)
                            a += cam_entry_term(rp, en);
                    };
                    if (own_lo)
                        add_rows(entry, acc);
                    if (own_hi)
                        add_rows(entry_hi, acc_hi);
                    total += rows;
                }
            }
            __syncthreads(); // (the buffer of round r - 1 is rewritten by round r + 1)
        }
        if (own_lo)
            normal[lane] = acc;
        if (own_hi)
            normal[lane + 64] = acc_hi;
        if (threadIdx.x == 0)
            s_count = total;
        __syncthreads();
    };

    // The same pass for ONE refined camera parameter (K = 7: pose + focal length, the LO and bundle of ransac_pnpf): measured with a
    // cycle counter in the kernel, the consumer of the general form spends 91 cycles per row (five LDS reads and the products of its
    // entry, next to three producer wavefronts writing rows of 31 doubles) and the producers wait for it four fifths of the pass.
    // Here the PRODUCERS form the 35 entry terms of their correspondence - the products cam_entry_term would form, in its operand
    // order - and store them column-major, [entry][row]; a third is padded with zero rows to 64 (the lanes without a row write
    // them: x + 0.0 = x), so the consumer lane of an entry adds a third with ONE inline-asm chain (pl_lm_chain.inc: 11.7 cycles
    // per row).  35 x 194 doubles per buffer.
    constexpr int kTermCols = 35, kTermStride = kProd + 2;
    auto jacobian_pass_m1 = [&](const double *p, const CameraParams &camera) {
        rotation_of(p);
        const Loss loss = ctl.loss;
        const CameraParams cam = camera;
        const int c6 = 6 + s_idx[0]; // the refined parameter's column of the row
        double acc = 0.0;
        uint32_t total = 0; // (consumer)
        for (uint32_t r = 0; r <= rounds; ++r) {
            if (wave > 0) {
                if (r < rounds) {
                    const uint32_t i = r * (uint32_t)kProd + (uint32_t)(pw * 64 + lane);
                    double row[kCamRow];
                    bool kept = false;
                    if (i < pts.n && !(mask && !mask[i]))
                        kept = abs_cam_row(p, s_R, cam, loss, pts.a[0][i] * pscale, pts.a[1][i] * pscale, pts.a[2][i], pts.a[3][i],
                                           pts.a[4][i], row);
                    const uint64_t b = __builtin_amdgcn_ballot_w64(kept);
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                    const uint32_t rows = (uint32_t)__popcll(b);
                    if (lane == 0)
                        s_wcnt[r & 1u][wave] = rows;
                    // kept rows in ascending order at 0 .. rows - 1, zero rows behind them
                    const uint32_t pos = kept ? below : rows + ((uint32_t)lane - below);
                    double *dst = s_rows + (size_t)(r & 1u) * kTermCols * kTermStride + (size_t)pw * 64 + pos;
                    if (kept) {
                        double J0[7], J1[7];
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            J0[k] = row[3 + k], J1[k] = row[3 + kCamMaxK + k];
                        J0[6] = row[3 + 6], J1[6] = row[3 + kCamMaxK + 6];
#pragma unroll
                        for (int m = 1; m < kCamMaxParams; ++m) { // (c6 is uniform)
                            J0[6] = c6 == 6 + m ? row[3 + 6 + m] : J0[6];
                            J1[6] = c6 == 6 + m ? row[3 + kCamMaxK + 6 + m] : J1[6];
                        }
                        const double w = row[0], wr0 = row[1], wr1 = row[2];
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < 7; ++a)
#pragma unroll
                            for (int c = 0; c <= a; ++c, ++e) { // cam_entry_term, triangle entry (a, c)
                                const double t = J0[a] * J0[c] + J1[a] * J1[c];
                                dst[(size_t)e * kTermStride] = w * t;
                            }
#pragma unroll
                        for (int a = 0; a < 7; ++a, ++e) { // gradient entry a
                            const double t = J0[a] * wr0 + J1[a] * wr1;
                            dst[(size_t)e * kTermStride] = 1.0 * t;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < kTermCols; ++e)
                            dst[(size_t)e * kTermStride] = 0.0;
                    }
                }
            } else if (r > 0) {
                const uint32_t buf = (r - 1u) & 1u;
                if (lane < kTermCols) {
#pragma unroll 1
                    for (int w = 0; w < kCamWaves - 1; ++w) { // the three thirds in order
                        const uint32_t addr = (uint32_t)(uintptr_t)(s_rows + (size_t)buf * kTermCols * kTermStride + (size_t)lane * kTermStride + (size_t)w * 64);
                        PL_LM_CHAIN64(acc, addr);
                    }
                }
                for (int w = 1; w < kCamWaves; ++w)
                    total += s_wcnt[buf][w];
            }
            __syncthreads(); // (the buffer of round r - 1 is rewritten by round r + 1)
        }
        if (own_lo)
            normal[lane] = acc;
        if (threadIdx.x == 0)
            s_count = total;
        __syncthreads();
    };

    cost_pass(cur, cam_cur);
    if (threadIdx.x == 0)
        lm_begin(ctl, T.opt, s_racc, s_count);
    __syncthreads();

    while (!ctl.done) {
        const bool fresh = ctl.rejac != 0;
        if (fresh) {
            if (M == 1)
                jacobian_pass_m1(cur, cam_cur);
            else
                jacobian_pass(cur, cam_cur);
        }
        if (threadIdx.x == 0) {
            lm_solve_k(K, ctl, normal, fresh, s_count);
            if (!ctl.done)
                abs_cam_step(cur, cam_cur, ctl.sol, s_idx, M, trial, cam_trial);
        }
        __syncthreads();
        if (ctl.done)
            break;
        cost_pass(trial, cam_trial);
        if (threadIdx.x == 0) {
            if (lm_update_k(K, ctl, normal, s_racc, s_count)) {
                for (int i = 0; i < kParamDoubles; ++i)
                    cur[i] = trial[i];
                cam_cur = cam_trial;
            }
        }
        __syncthreads();
    }

    if (threadIdx.x == 0) {
        for (int i = 0; i < kParamDoubles; ++i)
            Tout.params[i] = cur[i];
        Tout.cam = cam_cur;
        Tout.iterations = ctl.iterations;
        Tout.skipped = 0u;
        Tout.cost = ctl.cost;
        Tout.initial_cost = ctl.initial_cost;
        if (T.record_out)
            record_from_lm_params(EST_ABS, cur, T.record_out);
    }
}

hipError_t launch_lm_cam(LMTask *tasks, uint32_t num_tasks, hipStream_t stream) {
    if (num_tasks == 0)
        return hipSuccess;
    constexpr size_t bytes = sizeof(double) * 2 * std::max(kProd * kCamRow, 35 * (kProd + 2)); // 109 KB: two buffers of rows resp. of entry terms (one refined parameter)
    static std::atomic<int> prepared{0};
    if (!prepared.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lm_cam), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes);
        if (e != hipSuccess)
            return e;
        prepared.store(1, std::memory_order_release);
    }
    k_lm_cam<<<dim3(num_tasks), dim3(kCamThreads), bytes, stream>>>(tasks);
    return hipGetLastError();
}

} // namespace pl
