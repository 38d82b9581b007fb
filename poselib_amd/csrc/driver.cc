// poselib_amd — host driver behind the C-ABI (include/poselib_amd.h).
//
// What runs where
//   device : sampler positions (k_sample_delta / k_sample_orbit), sample draw + minimal solve (k_generate, k_rel_*),
//            hypothesis scoring (k_score_mfma / k_score_queue), the running-best scan (k_finalize2 / k_records), the
//            decision-relevant scores in the reference's summation order (k_score_seq), all LM refinements
//            (k_lm / k_lm2), final inlier masks (k_mask), the per-point pre-processing of the front-ends (k_prepare).
//   host   : the sequential bookkeeping of LO-RANSAC (RansacRun), replayed over the device results so that the outcome
//            equals the reference's single-threaded loop (PoseLib/robust/ransac_impl.h:106-201):
//              pass 1  the candidate list of a batch - hypotheses that improve the running (inlier count, MSAC score)
//                      in (iteration, model) order - through the exact rule; best_minimal_* depends on minimal models
//                      only (:113-123), so the iterations that trigger LO and their seeds are known without LO results;
//              device  all triggered LOs of the batch run as ONE batched LM launch, then are re-scored;
//              pass 2  replay stats.model_score / best_model / dynamic_max_iter and the stop rule (:182)
//                      iteration by iteration; iterations evaluated past the stop are discarded.
//            plus, of the front-ends (PoseLib/robust.cc:36-126, 242-314, 544-594, 712-757), the two order-dependent
//            reductions of normalize_points, threshold rescaling and de-normalisation; PROSAC's serial sample schedule.
// There is no CPU fallback for any device stage: without a HIP device the entry points fail.
#include "../../include/poselib_amd.h"
#include "pl_focal.h"
#include "pl_kernels.h"
#include "pl_refine_cam.h"
#include "pl_sfocal.h"
#include "pl_solver_6ptf.h"
#include "pl_solver_p35pf.h"
#include "pl_sampler.h"
#include "pl_svd3.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace pl;

namespace {

thread_local std::string g_err;

int fail(int code, const char *what, hipError_t e = hipSuccess) {
    g_err = what;
    if (e != hipSuccess) {
        g_err += ": ";
        g_err += hipGetErrorString(e);
    }
    return code;
}

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess)                                                                                          \
            return fail(PL_ERR_HIP, #expr, _e);                                                                        \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap)
            return hipSuccess;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess)
            cap = want;
        return e;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};
struct HostBuf { // pinned and mapped: kernels write their small results here themselves (no copy dispatch); the
                 // host reads them after the stream synchronisation that follows
    void *p = nullptr;
    void *dp = nullptr; // the same memory as the device addresses it
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap)
            return hipSuccess;
        if (p)
            (void)hipHostFree(p);
        p = dp = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped);
        if (e != hipSuccess)
            return e;
        e = hipHostGetDevicePointer(&dp, p, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(p);
            p = dp = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
    template <typename T> T *dev() const { return static_cast<T *>(dp); }
};

struct Context {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // batch scratch
    DevBuf positions, samples, shadow16, lm2_states, lm2_partials, raw_a, raw_b, pts_arena, absmax, models, num_models, slots, num_hyp, part_count, part_score, count, score;
    DevBuf shadow, compact64;
    DevBuf offsets, ctl, blk_best, rec_meta, rec_models, delta, flags;
    DevBuf gen_stage; // workspace of the staged 5-point generator
    DevBuf iota; // iota[i] = i: a device-resident "number of hypotheses" for launches whose count the host knows
    DevBuf lm_tasks, lm_records, gather_idx, gather_out, mask, lm_scratch, tmp_model, solve_in, solve_out, solve_cnt, focal_stage;
    HostBuf h_rec_meta, h_focal, h_in;
    const double *raw_src_a = nullptr, *raw_src_b = nullptr; // where the raw correspondences of the last make_problem_prepared live (device-visible)
    HostBuf h_absmax, h_positions, h_num_models, h_count, h_score, h_tasks, h_gather_idx, h_gather_out, h_mask,
        h_small, h_models;
    HostBuf h_flag;        // completion flag of wait_stream(): the stream writes a sequence number, the host spins on it
    uint32_t flag_seq = 0;
};

constexpr uint32_t kIotaEntries = 4097;
thread_local Context *g_ctx = nullptr;

thread_local int g_requested_device = 0;
thread_local const pl_shard *g_shard = nullptr; // set around ransac_core by pl_ransac_run_sharded

int get_context(Context **out) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(PL_ERR_NO_DEVICE, "no HIP device available (poselib_amd has no CPU fallback)", e);
    if (g_ctx && g_ctx->device == g_requested_device) {
        HIP_TRY(hipSetDevice(g_ctx->device));
        *out = g_ctx;
        return PL_OK;
    }
    if (g_requested_device < 0 || g_requested_device >= ndev)
        return fail(PL_ERR_INVALID, "device index out of range");
    // (a context bound to another device is simply leaked for the lifetime of the thread)
    {
        // the kernels are built for gfx950 only (k_sample_orbit alone keeps ~116 KB of static LDS): say so instead of
        // failing with "invalid device function" at the first launch
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, g_requested_device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            static thread_local std::string msg;
            msg = std::string("device architecture ") + prop.gcnArchName +
                  " is not supported: poselib_amd is built for gfx950 (MI355X) only";
            return fail(PL_ERR_UNSUPPORTED, msg.c_str());
        }
    }
    struct Guard { // releases a half-built context on the failure paths
        Context *c;
        ~Guard() {
            if (!c)
                return;
            if (c->ev0)
                (void)hipEventDestroy(c->ev0);
            if (c->ev1)
                (void)hipEventDestroy(c->ev1);
            if (c->stream)
                (void)hipStreamDestroy(c->stream);
            if (c->iota.p)
                (void)hipFree(c->iota.p);
            delete c;
        }
    } guard{new Context()};
    Context *c = guard.c;
    c->device = g_requested_device;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    {
        std::vector<uint32_t> iota(kIotaEntries);
        for (uint32_t i = 0; i < kIotaEntries; ++i)
            iota[i] = i;
        HIP_TRY(c->iota.ensure(sizeof(uint32_t) * kIotaEntries));
        HIP_TRY(hipMemcpy(c->iota.p, iota.data(), sizeof(uint32_t) * kIotaEntries, hipMemcpyHostToDevice));
    }
    if (c->h_flag.ensure(64) == hipSuccess)
        *c->h_flag.as<uint32_t>() = 0;
    else
        (void)hipGetLastError(); // (wait_stream falls back to hipStreamSynchronize)
    guard.c = nullptr;
    g_ctx = c;
    *out = c;
    return PL_OK;
}

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and kernels of streams
// that share a queue run one after the other.  The batch entry points keep 8 ... 16 launch chains in flight, one stream each: with
// the default, three of eight streams shared a queue and ran at half the rate of the two that had one to themselves
// (profiles/r04_batch_chain_4_queues.md; pl_estimate_batch 59.8 k -> 68 k problems/s).  The variable is read when the runtime
// initialises, i.e. at the process's first HIP call, and it belongs to the HOST APPLICATION: this library does not touch the
// process environment (round 4 set it from a constructor).  poselib_amd/_lib.py (the Python binding), bench.py and
// the reference-side binding's documentation set / recommend GPU_MAX_HW_QUEUES=16.

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Wait for everything enqueued on the context's stream.  Default: hipStreamSynchronize.  POSELIB_AMD_SPIN_SYNC=1: the
// stream itself writes a sequence number into pinned host memory behind the work (hipStreamWriteValue32: ordered like a
// kernel) and the host spins on that word (after 2 ms without progress it yields between polls, after 5 s it falls back
// to hipStreamSynchronize, which also reports a device fault instead of spinning for ever).  Measured on MI355X: one
// problem at a time 0.768 -> 0.743 ms per 100000-iteration P3P problem (three synchronisations each), but -13 % on the
// grouped batch (8 host threads) and -1 % with 16 problems in flight - the runtime's own wait already polls - so it is
// opt-in for latency-bound single-problem use.
// POSELIB_AMD_GROUP_TIMING=1 (diagnostic): pl_estimate_batch reports where its workers' time went - waiting for the device,
// in the per-item preparation, in fallback items - on stderr
std::atomic<uint64_t> g_t_wait_ns{0}, g_t_group_ns{0}, g_t_prep_ns{0}, g_t_fallback_ns{0}, g_n_waits{0}, g_n_fallback{0};
const bool g_group_timing = std::getenv("POSELIB_AMD_GROUP_TIMING") != nullptr;
hipError_t wait_stream_impl(Context *c);
// host work a worker of pl_estimate_batch does INSTEAD of idling at its next wait (staging the next group's inputs); one shot
thread_local std::function<void()> g_wait_hook;
thread_local std::vector<double> g_wait_marks; // (diagnostic) end of every wait of this worker's current job, seconds
hipError_t wait_stream(Context *c) {
    if (g_wait_hook) {
        std::function<void()> h;
        h.swap(g_wait_hook);
        h();
    }
    if (!g_group_timing)
        return wait_stream_impl(c);
    const double t0 = now_s();
    const hipError_t e = wait_stream_impl(c);
    const double t1 = now_s();
    g_t_wait_ns += (uint64_t)((t1 - t0) * 1e9);
    ++g_n_waits;
    g_wait_marks.push_back(t0);
    g_wait_marks.push_back(t1);
    return e;
}
hipError_t wait_stream_impl(Context *c) {
    static const bool spin = std::getenv("POSELIB_AMD_SPIN_SYNC") != nullptr;
    if (!spin || !c->h_flag.p)
        return hipStreamSynchronize(c->stream);
    const uint32_t seq = ++c->flag_seq;
    hipError_t e = hipStreamWriteValue32(c->stream, c->h_flag.dp, seq, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return hipStreamSynchronize(c->stream);
    }
    volatile uint32_t *flag = c->h_flag.as<volatile uint32_t>();
    const double t0 = now_s();
    for (uint64_t spins = 0;; ++spins) {
        if (*flag == seq) {
            std::atomic_thread_fence(std::memory_order_acquire); // (what the kernels wrote to pinned memory is read after this)
            return hipSuccess;
        }
        if ((spins & 0xfff) == 0xfff) {
            const double dt = now_s() - t0;
            if (dt > 5.0)
                return hipStreamSynchronize(c->stream);
            if (dt > 2e-3)
                std::this_thread::yield();
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

} // namespace

struct pl_problem {
    int kind;
    int device;
    uint32_t n;
    double *d_pts; // SoA block: nd arrays of n doubles
    PointSet ps;
    bool borrowed = false; // d_pts lives in the calling thread's context arena (one-shot front-ends): not freed
};

namespace {

LMOptions to_lm(const pl_bundle_options &b) {
    LMOptions o;
    o.max_iterations = (uint32_t)std::min<uint64_t>(b.max_iterations, 0xffffffffu);
    o.loss_type = b.loss_type;
    o.lambda_update = b.lambda_update;
    o.damping = b.damping;
    o.loss_scale = b.loss_scale;
    o.gradient_tol = b.gradient_tol;
    o.step_tol = b.step_tol;
    o.relative_cost_tol = b.relative_cost_tol;
    o.initial_lambda = b.initial_lambda;
    o.min_lambda = b.min_lambda;
    o.max_lambda = b.max_lambda;
    o.lambda_factor = b.lambda_factor;
    return o;
}
// bundle.refine_* as CamRefineFlags, reduced to the flags that select at least one parameter of the model (0: the pose alone,
// k_lm's case)
int active_cam_flags(int model_id, const pl_bundle_options &b) {
    const int flags = (b.refine_focal_length ? CAM_REFINE_FOCAL : 0) | (b.refine_principal_point ? CAM_REFINE_PRINCIPAL : 0) |
                      (b.refine_extra_params ? CAM_REFINE_EXTRA : 0);
    int idx[kCamMaxParams];
    return camera_refinement_idx(model_id, flags, idx) > 0 ? flags : 0;
}
LMOptions lo_options(double max_error) { // estimators/absolute_pose.cc:61-64 (identical in the four estimators)
    pl_bundle_options b;
    pl_default_bundle_options(&b);
    b.loss_type = LOSS_TRUNCATED;
    b.loss_scale = max_error;
    b.max_iterations = 25;
    return to_lm(b);
}
CameraParams to_cam(const pl_camera *c) {
    CameraParams r;
    std::memset(&r, 0, sizeof(r));
    if (!c) {
        r.model_id = CAM_NULL;
        return r;
    }
    r.model_id = c->model_id;
    r.num_params = c->num_params;
    for (int i = 0; i < 12; ++i)
        r.p[i] = c->params[i];
    return r;
}
bool camera_supported(const pl_camera *c) {
    return c->model_id == CAM_NULL || (c->model_id == CAM_SIMPLE_PINHOLE && c->num_params >= 3) ||
           (c->model_id == CAM_PINHOLE && c->num_params >= 4) || (c->model_id == CAM_OPENCV && c->num_params >= 8);
}
double camera_focal(const pl_camera *c) { // misc/camera_models.cc:304-323
    if (c->num_params == 0)
        return 1.0;
    switch (c->model_id) {
    case CAM_SIMPLE_PINHOLE:
        return 0.0 + c->params[0] / 1;
    case CAM_PINHOLE:
    case CAM_OPENCV:
        return 0.0 + c->params[0] / 2 + c->params[1] / 2;
    default:
        return 1.0;
    }
}
void camera_set_focal(pl_camera *c, double f) { // misc/camera_models.cc:96-107 over the model's focal_idx
    if (c->model_id == CAM_SIMPLE_PINHOLE && c->num_params >= 1) {
        c->params[0] = f;
    } else if ((c->model_id == CAM_PINHOLE || c->model_id == CAM_OPENCV) && c->num_params >= 2) {
        c->params[0] = f;
        c->params[1] = f;
    }
}
void camera_rescale(CameraParams &c, double s) { // misc/camera_models.cc:432-455
    if (c.num_params == 0)
        return;
    if (c.model_id == CAM_SIMPLE_PINHOLE) {
        c.p[0] *= s, c.p[1] *= s, c.p[2] *= s;
    } else if (c.model_id == CAM_PINHOLE || c.model_id == CAM_OPENCV) {
        for (int i = 0; i < 4; ++i)
            c.p[i] *= s;
    }
}

// ---- model <-> parameter block <-> record conversions (host side, IEEE add/mul only) ----
// `model` is the user-facing representation: pose = 7 doubles (q,t); matrix = 9 doubles ROW-major here.
void params_from_record(int kind, const double *rec, double *params) {
    for (int i = 0; i < kParamDoubles; ++i)
        params[i] = 0.0;
    if (kind == EST_ABS || kind == EST_REL) {
        for (int i = 0; i < 7; ++i)
            params[i] = rec[i];
    } else {
        for (int i = 0; i < 9; ++i)
            params[i] = rec[kMatOff + i];
    }
}

// Parameter block that enters k_lm for a model given as record.
void lm_params_from_record(int kind, const double *rec, double *params) {
    params_from_record(kind, rec, params);
    if (kind == EST_FUND) { // factorise F = U diag(1, sigma, 0) V^T  (bundle.cc:318-319)
        Mat3 F, U, V;
        for (int i = 0; i < 9; ++i)
            F.m[i] = rec[kMatOff + i];
        double s[3];
        svd3(F, U, s, V);
        if (det3(U) < 0)
            for (int i = 0; i < 9; ++i)
                U.m[i] = -U.m[i];
        if (det3(V) < 0)
            for (int i = 0; i < 9; ++i)
                V.m[i] = -V.m[i];
        const Quat qU = rotmat_to_quat(U), qV = rotmat_to_quat(V);
        for (int i = 0; i < kParamDoubles; ++i)
            params[i] = 0.0;
        params[0] = qU.w, params[1] = qU.x, params[2] = qU.y, params[3] = qU.z;
        params[4] = qV.w, params[5] = qV.x, params[6] = qV.y, params[7] = qV.z;
        params[8] = s[1] / s[0];
    }
}
void identity_record(int kind, double *rec) {
    if (kind == EST_ABS || kind == EST_REL) {
        Quat q;
        q.w = 1.0, q.x = q.y = q.z = 0.0;
        store_pose_model_q(rec, q, v3(0, 0, 0), kind == EST_REL);
    } else {
        Mat3 I;
        for (int i = 0; i < 9; ++i)
            I.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
        store_matrix_model(rec, I);
    }
}

// ---- iteration bound (ransac_impl.h:43-73) ----
double prob_all_inliers(uint64_t inl, uint64_t N, uint64_t K) {
    if (K == 0)
        return 1.0;
    if (inl < K || N < K)
        return 0.0;
    double p = 1.0;
    for (uint64_t i = 0; i < K; ++i)
        p *= static_cast<double>(inl - i) / static_cast<double>(N - i);
    return p;
}
uint64_t dynamic_max_iter(uint64_t inl, uint64_t N, uint64_t K, double log_fail, double mult, uint64_t min_it,
                          uint64_t max_it) {
    const double p = prob_all_inliers(inl, N, K);
    if (p >= 0.9999)
        return min_it;
    if (p <= 0.0001)
        return max_it;
    // The reference converts the double to size_t with a plain static_cast (ransac_impl.h:70-71), which is undefined
    // for values that do not fit - success_prob = 1 gives log(0) = -inf and hence +inf here.  Its de-facto behaviour
    // (gcc, x86-64: cvttsd2si on x - 2^63 and a sign-bit flip for x >= 2^63) is spelled out so that it does not depend
    // on the compiler of this file: +inf and everything >= 2^64 become 0, NaN becomes 2^63.
    const double v = std::ceil(log_fail / std::log(1.0 - p) * mult);
    const double two63 = 9223372036854775808.0;
    auto cvttsd2si = [&](double x) -> int64_t {
        return (x >= -two63 && x < two63) ? static_cast<int64_t>(x) : std::numeric_limits<int64_t>::min();
    };
    const uint64_t n = (v >= two63) ? (static_cast<uint64_t>(cvttsd2si(v - two63)) ^ 0x8000000000000000ull)
                                    : static_cast<uint64_t>(cvttsd2si(v));
    return std::max(min_it, std::min(max_it, n));
}

// Host replay of the sampler to find where each iteration's draws start (integer work, a few ns per
// draw): out[i] = draws consumed before iteration i, RELATIVE to `pos` (like the device's k_sample_orbit; the
// generators add GenerateArgs.pos_base).  Returns the absolute draw position after `count` iterations.
template <int K> uint64_t sample_positions(uint64_t seed, uint64_t pos, uint64_t N, uint32_t count, uint32_t *out) {
    uint32_t idx[K];
    const uint64_t pos0 = pos;
    for (uint32_t i = 0; i < count; ++i) {
        out[i] = (uint32_t)(pos - pos0);
        pos += draw_sample<K>(seed, pos, N, idx);
    }
    return pos;
}
uint64_t sample_positions_k(int K, uint64_t seed, uint64_t pos, uint64_t N, uint32_t count, uint32_t *out) {
    switch (K) {
    case 3:
        return sample_positions<3>(seed, pos, N, count, out);
    case 4:
        return sample_positions<4>(seed, pos, N, count, out);
    case 5:
        return sample_positions<5>(seed, pos, N, count, out);
    default:
        return sample_positions<7>(seed, pos, N, count, out);
    }
}

// PROSAC sampling (PoseLib/robust/sampling.cc:85-136): the size of the subset the sample is drawn from follows a
// serial recurrence over the iteration counter, so the samples of a batch are drawn on the host - one splitmix64
// draw (pl_sampler.h draw_at) per index - and handed to k_generate explicitly.  After max_prosac_iterations the
// sampler is the uniform one again (sampling.cc:101-102).
// (ProsacSampler: pl_sampler.h)

// Parameters of the scoring kernels' conservative fp32 pre-filters (pl_prefilter.h).  Every value is rounded UP;
// POSELIB_AMD_NO_PREFILTER=1 disables the filters (exact evaluation of every point).
void set_prefilter(ScoreArgs &sa, const pl_problem *p, double thr2) {
    static const bool disabled = std::getenv("POSELIB_AMD_NO_PREFILTER") != nullptr;
    sa.pf = make_prefilter_args(p->kind, thr2, p->ps.xy_absmax);
    if (disabled)
        sa.pf.enabled = 0;
}

// Score `nrec` model records that already sit in device memory at `d_records`, in the reference's summation order
// (k_score_seq).  Results land in c->count / c->score and in the pinned h_count / h_score buffers after the caller
// synchronises.
int enqueue_score_records(Context *c, const pl_problem *p, const double *d_records, uint32_t nrec, double thr2,
                          bool time_it) {
    (void)time_it;
    HIP_TRY(c->num_hyp.ensure(sizeof(uint32_t)));
    HIP_TRY(c->count.ensure(sizeof(uint32_t) * nrec));
    HIP_TRY(c->score.ensure(sizeof(double) * nrec));
    HIP_TRY(c->h_count.ensure(sizeof(uint32_t) * nrec));
    HIP_TRY(c->h_score.ensure(sizeof(double) * nrec));
    const bool counted = nrec < kIotaEntries; // the count is read from the device-resident table: no upload
    if (!counted)
        HIP_TRY(hipMemcpyAsync(c->num_hyp.p, &nrec, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    SeqScoreArgs sa;
    sa.pts = p->ps;
    sa.models = d_records;
    sa.cand = nullptr;
    sa.num = counted ? c->iota.as<uint32_t>() + nrec : c->num_hyp.as<uint32_t>();
    sa.cap = nrec;
    sa.thr2 = thr2;
    sa.count = c->count.as<uint32_t>();
    sa.score = c->score.as<double>();
    sa.host_count = c->h_count.dev<uint32_t>(); // the kernel writes the pinned copies itself
    sa.host_score = c->h_score.dev<double>();
    sa.host_cand = nullptr;
    sa.host_cap = 0;
    HIP_TRY(launch_score_seq(p->kind, sa, c->stream));
    return PL_OK;
}

constexpr uint32_t kRecordCap = 1024;  // device record list capacity (overflow -> host scan fallback)
constexpr uint32_t kRecordFirst = 64;  // records fetched together with the control block

struct RefineJob {
    double record_in[kModelStride]; // seed model
    LMOptions opt;
    CameraParams cam;
    double point_scale = 1.0;
    double prefilter_thr2 = 0.0;
    const uint8_t *d_mask = nullptr;
    int cam_flags = 0; // absolute pose: CamRefineFlags, the intrinsics refined with the pose (k_lm_cam)
    // outputs
    CameraParams cam_out;
    double record_out[kModelStride];
    double params_out[kParamDoubles];
    uint32_t count = 0;
    double score = 0;
    bool skipped = false;
};

// Optional tail of a single-job refinement: pick "refined if its score beats the incumbent's, else the incumbent"
// on the device (ransac_impl.h:190-198) and compute that model's inlier mask, all behind the same synchronisation.
struct MaskTail {
    double incumbent_score;      // stats.model_score
    const double *incumbent_rec; // host record of the incumbent (best model so far)
    double thr2;
    uint8_t *host_mask;          // N bytes or nullptr
};

// Runs all jobs as one batched LM launch; the refined parameters become records ON THE DEVICE (k_task_records),
// which are (optionally) re-scored with threshold thr2, and (optionally, single job) followed by the mask tail.
// One stream synchronisation for the whole chain.
int run_refinements(Context *c, const pl_problem *p, std::vector<RefineJob> &jobs, bool rescore, double thr2,
                    const MaskTail *tail = nullptr) {
    const uint32_t nj = (uint32_t)jobs.size();
    if (nj == 0)
        return PL_OK;
    // one staging block in pinned, device-mapped host memory: [LMTask x nj][seed records x nj (+ the incumbent's record)].
    // k_lm fetches its task from there itself (one cooperative read over the bus) and writes the outputs back, so the
    // single-workgroup path needs no upload dispatch; the multi-launch k_lm2 path reads its tasks every launch and keeps
    // a device copy.
    static_assert(sizeof(LMTask) % sizeof(double) == 0, "records follow the tasks");
    const size_t stage_bytes = sizeof(LMTask) * nj + sizeof(double) * kModelStride * (nj + 1);
    HIP_TRY(c->h_tasks.ensure(stage_bytes));
    HIP_TRY(c->lm_scratch.ensure((size_t)p->n * nj + 16));
    HIP_TRY(c->lm_records.ensure(sizeof(double) * kModelStride * nj));
    // Large point sets with 6..8 parameters saturate the single CU k_lm gives a task; k_lm2 spreads every task over
    // several workgroups (one launch per LM iteration; only for short, bounded runs - the LO: 25 iterations - since it
    // costs max_iterations + 2 launches whatever the iteration count turns out to be).  Measured on MI355X at N = 10^4:
    // homography problems 2.2x, fundamental 1.5x shorter when the device serves one problem at a time; with many
    // problems in flight the long single-CU kernels overlap anyway and the extra launches cost throughput.  The two
    // kernels sum the normal equations in different orders (results agree to rounding, not to the bit), so the choice
    // must not depend on load or on the entry point: it is a process-wide setting, OFF by default - the grouped launches
    // of pl_estimate_batch refine with k_lm, and a problem gives the same bits whichever way it is submitted -,
    // POSELIB_AMD_LATENCY_MODE=1 switches it on for single large two-view problems.
    static const bool latency_mode = [] {
        const char *e = std::getenv("POSELIB_AMD_LATENCY_MODE");
        return e && e[0] == '1';
    }();
    uint32_t max_it = 0;
    bool same_it = true;
    for (uint32_t j = 0; j < nj; ++j) {
        same_it = same_it && (j == 0 || jobs[j].opt.max_iterations == max_it);
        max_it = std::max(max_it, jobs[j].opt.max_iterations);
    }
    const uint32_t lm2_min_points = (p->kind == EST_ABS) ? 8192u : 2560u;
    bool with_camera = false;
    for (uint32_t j = 0; j < nj; ++j)
        with_camera = with_camera || jobs[j].cam_flags != 0;
    if (with_camera && p->kind != EST_ABS)
        return fail(PL_ERR_INVALID, "camera intrinsics are refined with absolute poses only");
    const bool use_lm2 = !with_camera && latency_mode && p->kind != EST_REL && !lm_sums_ordered(p->kind) && same_it && max_it >= 1 && max_it <= 32 && p->n >= lm2_min_points;
    if (use_lm2)
        HIP_TRY(c->lm_tasks.ensure(stage_bytes));
    LMTask *ht = c->h_tasks.as<LMTask>();
    double *hr = reinterpret_cast<double *>(ht + nj);
    // where the kernels see the staging block: the device copy (k_lm2) or the pinned block itself (k_lm)
    LMTask *d_tasks = use_lm2 ? c->lm_tasks.as<LMTask>() : c->h_tasks.dev<LMTask>();
    double *d_records_in = reinterpret_cast<double *>(d_tasks + nj);
    for (uint32_t j = 0; j < nj; ++j) {
        LMTask &t = ht[j];
        std::memset(&t, 0, sizeof(t));
        t.pts = p->ps;
        lm_params_from_record(p->kind, jobs[j].record_in, t.params);
        t.opt = jobs[j].opt;
        t.cam = jobs[j].cam;
        t.point_scale = jobs[j].point_scale;
        t.cam_flags = jobs[j].cam_flags;
        t.prefilter_thr2 = jobs[j].prefilter_thr2;
        t.mask = jobs[j].d_mask;
        t.scratch = c->lm_scratch.as<uint8_t>() + (size_t)j * p->n;
        t.record_in = d_records_in + (size_t)j * kModelStride;
        t.record_out = use_lm2 ? nullptr : c->lm_records.as<double>() + (size_t)j * kModelStride;
        std::memcpy(hr + (size_t)j * kModelStride, jobs[j].record_in, sizeof(double) * kModelStride);
    }
    if (tail)
        std::memcpy(hr + (size_t)nj * kModelStride, tail->incumbent_rec, sizeof(double) * kModelStride);
    if (use_lm2) {
        HIP_TRY(hipMemcpyAsync(c->lm_tasks.p, ht, stage_bytes, hipMemcpyHostToDevice, c->stream));
        const uint32_t slices = std::min<uint32_t>(16u, std::max<uint32_t>(2u, p->n / 512u));
        HIP_TRY(c->lm2_states.ensure(lm2_state_bytes(nj)));
        HIP_TRY(c->lm2_partials.ensure(lm2_partial_bytes(nj, slices)));
        HIP_TRY(launch_lm2(p->kind, p->ps, d_tasks, nj, slices, max_it, c->lm2_states.p, c->lm2_partials.as<double>(),
                           c->stream));
        // (k_task_records also writes the tasks' outputs into the pinned staging block: stream order puts that after the
        // upload above has read it)
        HIP_TRY(launch_task_records(p->kind, d_tasks, d_records_in, c->lm_records.as<double>(), nj,
                                    c->h_tasks.dev<LMTask>(), c->stream));
    } else if (with_camera) { // (tasks of such a call without flags: M = 0 is not a k_lm_cam case)
        for (uint32_t j = 0; j < nj; ++j)
            if (!jobs[j].cam_flags)
                return fail(PL_ERR_INVALID, "refinement jobs with and without intrinsics in one launch");
        HIP_TRY(launch_lm_cam(d_tasks, nj, c->stream));
    } else {
        HIP_TRY(launch_lm(p->kind, p->ps, d_tasks, nj, c->stream)); // outputs + refined records: written by k_lm itself
    }
    if (rescore || tail) {
        int rc = enqueue_score_records(c, p, c->lm_records.as<double>(), nj, tail ? tail->thr2 : thr2, false);
        if (rc != PL_OK)
            return rc;
    }
    if (tail) { // single job: choose refined / incumbent on the device, then the inlier mask of the choice
        HIP_TRY(c->mask.ensure(std::max<uint32_t>(p->n, 1u)));
        HIP_TRY(c->tmp_model.ensure(sizeof(double) * kModelStride));
        HIP_TRY(launch_select_record(c->score.as<double>(), tail->incumbent_score, c->lm_records.as<double>(),
                                     d_records_in + (size_t)nj * kModelStride,
                                     c->tmp_model.as<double>(), c->stream));
        HIP_TRY(c->h_mask.ensure(std::max<uint32_t>(p->n, 1u)));
        HIP_TRY(launch_mask(p->kind, p->ps, c->tmp_model.as<double>(), tail->thr2, c->mask.as<uint8_t>(),
                            tail->host_mask ? c->h_mask.dev<uint8_t>() : nullptr, c->stream));
    }
    HIP_TRY(wait_stream(c));
    if (tail && tail->host_mask && p->n) // the kernel wrote the pinned copy itself
        std::memcpy(tail->host_mask, c->h_mask.p, p->n);
    for (uint32_t j = 0; j < nj; ++j) {
        jobs[j].skipped = ht[j].skipped != 0;
        std::memcpy(jobs[j].params_out, ht[j].params, sizeof(double) * kParamDoubles);
        jobs[j].cam_out = ht[j].cam;
        if (jobs[j].skipped) // refinement not run: model unchanged (relative_pose.cc:75-77)
            std::memcpy(jobs[j].record_out, jobs[j].record_in, sizeof(double) * kModelStride);
        else
            record_from_lm_params(p->kind, ht[j].params, jobs[j].record_out); // same inline function as the device
        if (rescore || tail) {
            jobs[j].count = c->h_count.as<uint32_t>()[j];
            jobs[j].score = c->h_score.as<double>()[j];
        }
    }
    return PL_OK;
}

struct Improving { // a minimal hypothesis that improved best_minimal_* (candidate for best_model)
    uint32_t iter;   // absolute iteration index
    uint32_t slot;   // model record index inside the batch
    uint32_t count;
    double score;
    bool lo_seed;    // last improving hypothesis of its iteration
    uint32_t gather; // index into the gathered record array
    int job = -1;    // index of the refinement job when lo_seed
};

// One LO-RANSAC run (ransac_impl.h:157-201) as batches: `enqueue_batch` puts the whole device pipeline of a batch of
// iterations on the stream, `collect_improving` synchronises once and lists the hypotheses that improved the running
// best, `exchange_improving` (sharded runs only) merges the ranks' lists, `refine_and_replay` runs the triggered local
// optimisations as one batched launch and replays the sequential loop over the batch, event by event.
struct RansacRun {
    static constexpr int kRedoBatch = 1; // positive: not an error, the batch has to be evaluated again

    // what one batch hands from step to step
    struct Batch {
        uint32_t B = 0;                      // iterations of the batch (all ranks)
        uint32_t lo_g = 0, hi_g = 0, Bl = 0; // this rank's share: iterations [it + lo_g, it + hi_g)
        size_t hcap = 0;                     // hypothesis capacity of the share
        BatchCtl *d_ctl = nullptr;
        uint64_t pos_after = 0; // sampler draws consumed after the batch
        bool device_positions = false;
        ProsacSampler prosac_at_batch_start; // a repeated batch draws the same samples
        bool overflow = false;               // an iteration of the share produced more models than slots
        uint32_t H = 0, H_local = 0;         // hypotheses of the batch (sharded: over all ranks) / of the share
        uint64_t b0_inl = 0;                 // state of the sequential loop at the start of the batch
        double b0_score = 0;
        const double *h_rec = nullptr; // model records of `imps`, indexed by Improving::gather
        bool ctl_mirrored = false;     // k_score_seq has written the control block to pinned host memory
    };

    Context *c;
    const pl_problem *p;
    const pl_robust_options *o;
    double *best_record; // in/out
    uint8_t *inliers;
    pl_ransac_stats *st;
    const int kind;
    const uint32_t N;
    const int K;
    int MAXM; // record slots per iteration (5-pt: grown to 40 on demand)
    const pl_ransac_options &ro;
    const double thr2;
    const LMOptions lo_opt;
    CameraParams null_cam;
    // one problem across several devices (pl_ransac_run_sharded): this rank evaluates a contiguous share of every batch
    const pl_shard *sh;
    const uint32_t G, grank;
    struct WireHead {
        uint32_t n, gen_overflow, H, pad;
    };
    struct WireRec {
        uint32_t iter_off, count; // iteration relative to the batch start
        double score;
        double model[kModelStride];
    };
    static constexpr uint32_t kWireFirst = 32; // records that travel with the header
    std::vector<unsigned char> wire_send, wire_recv;
    std::vector<double> merged_models;
    // state of the sequential loop
    uint64_t best_min_inl = 0;
    double best_min_score = std::numeric_limits<double>::max();
    uint64_t dyn_max;
    const double log_fail;
    uint64_t it = 0;  // next iteration to evaluate == iterations replayed so far
    uint64_t pos = 0; // sampler draws consumed so far
    bool stopped = false;
    std::vector<Improving> imps;
    std::vector<RefineJob> jobs;
    std::vector<uint32_t> order;
    // diagnostics: POSELIB_AMD_HOST_BOOKKEEPING=1 (both), POSELIB_AMD_HOST_POSITIONS=1 (sampler positions walked on the
    // host), POSELIB_AMD_HOST_RECORDS=1 (improving hypotheses found by a host scan over all scores)
    const bool host_bookkeeping = std::getenv("POSELIB_AMD_HOST_BOOKKEEPING") != nullptr;
    const bool host_positions = host_bookkeeping || std::getenv("POSELIB_AMD_HOST_POSITIONS") != nullptr;
    const bool host_records = host_bookkeeping || std::getenv("POSELIB_AMD_HOST_RECORDS") != nullptr;
    bool force_host_positions = false;
    const bool prosac;
    ProsacSampler prosac_sampler;

    RansacRun(Context *c_, const pl_problem *p_, const pl_robust_options *o_, double *best_record_, uint8_t *inliers_,
              pl_ransac_stats *st_)
        : c(c_), p(p_), o(o_), best_record(best_record_), inliers(inliers_), st(st_), kind(p_->kind), N(p_->n),
          K(sample_size(p_->kind)), MAXM((p_->kind == EST_REL) ? 8 : max_models(p_->kind)), ro(o_->ransac),
          thr2(o_->max_error * o_->max_error), lo_opt(lo_options(o_->max_error)),
          sh((g_shard && g_shard->world > 1) ? g_shard : nullptr), G(sh ? (uint32_t)sh->world : 1u),
          grank(sh ? (uint32_t)sh->rank : 0u), dyn_max(o_->ransac.max_iterations),
          log_fail(std::log(1.0 - o_->ransac.success_prob)), prosac(o_->ransac.progressive_sampling != 0) {
        std::memset(&null_cam, 0, sizeof(null_cam));
        null_cam.model_id = CAM_NULL;
    }

    RefineJob make_lo_job(const double *rec) const {
        RefineJob j;
        std::memcpy(j.record_in, rec, sizeof(j.record_in));
        j.opt = lo_opt;
        j.cam = null_cam;
        j.point_scale = 1.0;
        j.prefilter_thr2 = (kind == EST_REL) ? 5 * thr2 : 0.0; // relative_pose.cc:70
        return j;
    }
    // A homography is a matrix up to scale AND sign, and two local optimisations that started from minimal models of opposite
    // sign end in the same optimum as H and -H with scores that agree to the last few bits: which of them `score < best` keeps
    // hangs on the bits of the refined models, i.e. on the order the normal equations are summed in.  Beyond 256 correspondences
    // the default sums run in tree order (models 1e-13 off the reference's), and in one of ~1000 default-option problems the
    // caller got -H (tests/parity_soak_estimate_batch.py 4000 11, item 1186; every other field identical).  A local
    // optimisation is a function of its seed - the minimal model, whose bits ARE the reference's - so when such a decision
    // comes up (opposite signs, scores within 2e-15: the tree order moves a converged model's score by an ulp or two) the two
    // refinements are repeated with the sums in the reference's order (k_lm_ordered) and the decision is taken on those
    // results, which are the reference's bit for bit.  Up to 4096 correspondences: there the ordered kernel costs 1.3 - 2 x the
    // tree kernel for the one or two tasks concerned (< 0.5 % of the default-option homography problems; with a bound of 1e-13,
    // 1.5 % of them, the synchronous launches inside the groups' replays cost a 512-problem batch call 17 %); at 10^4 it costs 6 x
    // and long runs meet such a pair about once each - that regime keeps the tree order (pl_set_lm_mode(1) pins it).
    double best_seed[kModelStride];                  // the minimal model the incumbent was refined from ...
    bool best_from_lo = false, best_exact = false;   // ... if it is a refined one / already refined in the reference's order
    static constexpr double kSignTieGap = 2e-15; // (9 ulps of the score: the tree order moves a converged model's score by an ulp or two)
    static constexpr uint32_t kSignTieMaxPoints = 4096;
    int resolve_sign_tie(RefineJob &job) {
        static const bool off = std::getenv("POSELIB_AMD_NO_SIGN_TIE") != nullptr; // (diagnostic: A/B of what the extra refinements cost)
        // (a sharded run: every rank holds the whole problem and replays the same decisions - each repeats the two refinements itself)
        if (off || kind != EST_HOM || N <= (uint32_t)kLMSeqPoints || N > kSignTieMaxPoints || lm_sums_ordered(EST_HOM) || job.skipped ||
            !(st->model_score < std::numeric_limits<double>::max()) ||
            !(std::fabs(job.score - st->model_score) <= kSignTieGap * std::fabs(st->model_score)))
            return PL_OK;
        double dot = 0;
        for (int i = 0; i < 9; ++i)
            dot += job.record_out[kMatOff + i] * best_record[kMatOff + i];
        if (!(dot < 0))
            return PL_OK;
        std::vector<RefineJob> exact{make_lo_job(job.record_in)};
        const bool both = best_from_lo && !best_exact;
        if (both)
            exact.push_back(make_lo_job(best_seed));
        set_lm_force_ordered(1);
        const int rc = run_refinements(c, p, exact, true, thr2);
        set_lm_force_ordered(0);
        if (rc != PL_OK)
            return rc;
        job.score = exact[0].score, job.count = exact[0].count, job.skipped = exact[0].skipped;
        std::memcpy(job.record_out, exact[0].record_out, sizeof(job.record_out));
        std::memcpy(job.params_out, exact[0].params_out, sizeof(job.params_out));
        if (both) {
            st->model_score = exact[1].score;
            st->num_inliers = exact[1].count;
            std::memcpy(best_record, exact[1].record_out, sizeof(double) * kModelStride);
            best_exact = true;
        }
        job_exact = true;
        return PL_OK;
    }
    bool job_exact = false;
    int after_lo(RefineJob &job) { // ransac_impl.h:138-153
        st->refinements++;
        job_exact = false;
        const int rc_tie = resolve_sign_tie(job);
        if (rc_tie != PL_OK)
            return rc_tie;
        if (job.score < st->model_score) {
            st->model_score = job.score;
            st->num_inliers = job.count;
            std::memcpy(best_record, job.record_out, sizeof(double) * kModelStride);
            std::memcpy(best_seed, job.record_in, sizeof(best_seed));
            best_from_lo = true;
            best_exact = job_exact;
        }
        st->inlier_ratio = static_cast<double>(st->num_inliers) / static_cast<double>(N);
        dyn_max = dynamic_max_iter(st->num_inliers, N, K, log_fail, ro.dyn_num_trials_mult, ro.min_iterations,
                                   ro.max_iterations);
        return PL_OK;
    }

    int score_initial_model() {
        HIP_TRY(c->tmp_model.ensure(sizeof(double) * kModelStride));
        HIP_TRY(hipMemcpyAsync(c->tmp_model.p, best_record, sizeof(double) * kModelStride, hipMemcpyHostToDevice,
                               c->stream));
        int rc = enqueue_score_records(c, p, c->tmp_model.as<double>(), 1, thr2, false);
        if (rc != PL_OK)
            return rc;
        HIP_TRY(wait_stream(c));
        const uint32_t cnt = c->h_count.as<uint32_t>()[0];
        const double sc = c->h_score.as<double>()[0];
        const bool more = cnt > best_min_inl, better = sc < best_min_score;
        if (more || better) {
            if (more)
                best_min_inl = cnt;
            if (better)
                best_min_score = sc;
            if (sc < st->model_score) {
                st->model_score = sc;
                st->num_inliers = cnt;
            }
            std::vector<RefineJob> jobs{make_lo_job(best_record)};
            rc = run_refinements(c, p, jobs, true, thr2);
            if (rc != PL_OK)
                return rc;
            rc = after_lo(jobs[0]);
            if (rc != PL_OK)
                return rc;
        }
        return PL_OK;
    }

    // ---- device: positions -> generate -> compact -> score -> finalize -> records, all on the stream ----
    int enqueue_batch(Batch &b) {
        const uint32_t B = b.B, lo_g = b.lo_g, hi_g = b.hi_g, Bl = b.Bl;
        (void)B, (void)lo_g, (void)hi_g, (void)Bl;
        // ---- device: positions -> generate -> compact -> score -> finalize -> records ----
        const size_t hcap = (size_t)std::max<uint32_t>(Bl, 1u) * MAXM;
        ScoreArgs sa;
        set_prefilter(sa, p, thr2);
        const bool prefilter = true; // compact hypothesis stream for the streaming scorer (all estimators)
        const bool on_mfma = score_uses_mfma(kind, N, sa.pf);
        const uint32_t chunks = score_chunks(kind, N, prefilter, on_mfma);
        if (prefilter) {
            HIP_TRY(c->shadow.ensure(sizeof(float) * 16 * hcap));
            HIP_TRY(c->compact64.ensure(sizeof(double) * kModelDoubles * hcap));
        }
        HIP_TRY(c->positions.ensure(sizeof(uint32_t) * B));
        HIP_TRY(c->models.ensure(sizeof(double) * kModelStride * hcap));
        HIP_TRY(c->num_models.ensure(sizeof(uint32_t) * std::max<uint32_t>(Bl, 1u)));
        HIP_TRY(c->slots.ensure(sizeof(uint32_t) * hcap));
        HIP_TRY(c->offsets.ensure(sizeof(uint32_t) * std::max<uint32_t>(Bl, 1u)));
        // control block + models per 1024 iterations (zeroed together, filled by the generator)
        // behind the control block, zeroed with it: models and NaN models per 1024 iterations (two tables of nblk + 1
        // entries), the scorer's work counters (one per chunk of correspondences)
        const uint32_t nblk = (Bl + 1023) / 1024;
        const size_t ctl_bytes = sizeof(BatchCtl) + sizeof(uint32_t) * (2 * (size_t)nblk + 2 + chunks);
        HIP_TRY(c->ctl.ensure(ctl_bytes));
        HIP_TRY(c->part_count.ensure(sizeof(uint32_t) * chunks * hcap));
        HIP_TRY(c->part_score.ensure(sizeof(double) * chunks * hcap));
        HIP_TRY(c->count.ensure(sizeof(uint32_t) * hcap));
        HIP_TRY(c->score.ensure(sizeof(double) * hcap));
        HIP_TRY(c->blk_best.ensure((sizeof(uint32_t) + sizeof(double)) * 256 + 64));
        HIP_TRY(c->rec_meta.ensure(sizeof(RecordMeta) * kRecordCap));
        HIP_TRY(c->rec_models.ensure(sizeof(double) * kModelStride * kRecordCap));
        HIP_TRY(c->h_small.ensure(sizeof(BatchCtl) + 64));
        HIP_TRY(c->h_rec_meta.ensure(sizeof(RecordMeta) * kRecordCap));
        HIP_TRY(c->h_gather_out.ensure(sizeof(double) * kModelStride * kRecordCap));
        BatchCtl *d_ctl = c->ctl.as<BatchCtl>();
        bool ctl_zeroed = false; // (the device sampler's first kernel clears the block; otherwise a memset does)
        uint32_t *const blk_tot = reinterpret_cast<uint32_t *>(d_ctl + 1);

        uint64_t pos_after = 0;
        bool device_positions = !host_positions && !force_host_positions && !prosac;
        const ProsacSampler prosac_at_batch_start = prosac_sampler; // a repeated batch draws the same samples
        static_assert(sizeof(BatchCtl) % 4 == 0, "zeroed as 32-bit words");
        if (prosac) {
            HIP_TRY(c->h_positions.ensure(sizeof(uint32_t) * (size_t)B * K));
            HIP_TRY(c->samples.ensure(sizeof(uint32_t) * (size_t)B * K));
            uint32_t *hs = c->h_positions.as<uint32_t>();
            for (uint32_t b = 0; b < B; ++b)
                prosac_sampler.generate(hs + (size_t)b * K);
            pos_after = prosac_sampler.pos;
            if (Bl)
                HIP_TRY(hipMemcpyAsync(c->samples.p, hs + (size_t)lo_g * K, sizeof(uint32_t) * (size_t)Bl * K,
                                       hipMemcpyHostToDevice, c->stream));
        }
        if (device_positions) {
            // window of draw positions to evaluate: expected draws per iteration (sum N/(N-i)) + slack
            double per_it = 0;
            for (int i = 0; i < K; ++i)
                per_it += static_cast<double>(N) / static_cast<double>(N - i);
            const uint64_t M64 = (uint64_t)(B * per_it * 1.05) + 8192;
            if (M64 > 0x7fffffffull || pos + M64 >= 0xffffffffull) {
                device_positions = false;
            } else {
                const uint32_t M = (uint32_t)M64;
                HIP_TRY(c->delta.ensure((size_t)M + 64));
                HIP_TRY(c->flags.ensure(sizeof(uint64_t) * ((size_t)M / 64 + 2))); // bitmap of redrawing positions
                HIP_TRY(launch_sample_positions(K, ro.seed, pos, N, B, M, c->delta.as<uint8_t>(),
                                                c->flags.as<uint64_t>(), c->positions.as<uint32_t>(), d_ctl,
                                                (uint32_t)(ctl_bytes / 4), c->stream));
                ctl_zeroed = true;
            }
        }
        if (!ctl_zeroed)
            HIP_TRY(hipMemsetAsync(d_ctl, 0, ctl_bytes, c->stream));
        if (!device_positions && !prosac) {
            HIP_TRY(c->h_positions.ensure(sizeof(uint32_t) * B));
            pos_after = sample_positions_k(K, ro.seed, pos, N, B, c->h_positions.as<uint32_t>());
            if (pos_after - pos >= 0xffffffffull)
                return fail(PL_ERR_UNSUPPORTED, "sampler draw window exceeds 32 bits");
            HIP_TRY(hipMemcpyAsync(c->positions.p, c->h_positions.p, sizeof(uint32_t) * B, hipMemcpyHostToDevice,
                                   c->stream));
        }
        if (Bl > 0) { // (a rank whose share of a short batch is empty only takes part in the exchange)
            GenerateArgs ga;
            ga.pts = p->ps;
            ga.seed = ro.seed;
            ga.pos_base = pos;
            ga.positions = c->positions.as<uint32_t>() + lo_g;
            ga.samples = prosac ? c->samples.as<uint32_t>() : nullptr;
            ga.num_iters = Bl;
            ga.slots_per_iter = (uint32_t)MAXM;
            ga.ctl = d_ctl;
            ga.models = c->models.as<double>();
            ga.num_models = c->num_models.as<uint32_t>();
            ga.real_focal_check = o->real_focal_check;
            ga.blk_tot = blk_tot;
            ga.blk_nan = blk_tot + nblk;
            if (const size_t sb = generate_stage_bytes(kind, Bl)) {
                HIP_TRY(c->gen_stage.ensure(sb));
                ga.stage = c->gen_stage.p;
            }
            HIP_TRY(launch_generate(kind, ga, c->stream));
            sa.pts = p->ps;
            sa.models = ga.models;
            sa.slots = c->slots.as<uint32_t>();
            sa.shadow16 = nullptr;
            Shadow16Params s16;
            if (on_mfma && kind == EST_ABS) { // fp16 operand blocks of the hypotheses for the matrix cores: built
                                              // in the same launch as the hypothesis-ordered copies
                HIP_TRY(c->shadow16.ensure((hcap + kAbs16Pad) * kAbs16Bytes));
                s16.out = c->shadow16.p;
                s16.g16 = sa.pf.g16, s16.c16 = sa.pf.c16, s16.thr = sa.pf.thr;
                sa.shadow16 = c->shadow16.p;
            } else if (on_mfma && kind == EST_HOM) { // homography: operands of k_score_mfmah (k_hom16)
                HIP_TRY(c->shadow16.ensure((hcap + kHom16Pad) * kHom16Bytes));
                s16.out = c->shadow16.p;
                s16.sampson = 2;
                s16.thr = sa.pf.h16;
                sa.shadow16 = c->shadow16.p;
            } else if (on_mfma) { // two-view: operands of the Sampson forms (k_sampson16 / k_score_mfma2)
                HIP_TRY(c->shadow16.ensure((hcap + kSampson16Pad) * kSampson16Bytes));
                s16.out = c->shadow16.p;
                s16.sampson = 1;
                sa.shadow16 = c->shadow16.p;
            }
            HIP_TRY(launch_compact2(ga.num_models, Bl, MAXM, blk_tot, true, c->slots.as<uint32_t>(),
                                    c->offsets.as<uint32_t>(), ga.models, prefilter ? c->shadow.as<float>() : nullptr,
                                    prefilter ? c->compact64.as<double>() : nullptr, d_ctl, s16, c->stream));
            sa.shadow = prefilter ? c->shadow.as<float>() : nullptr;
            sa.compact64 = prefilter ? c->compact64.as<double>() : nullptr;
            sa.num_hyp = &d_ctl->num_hyp;
            sa.hyp_capacity = (uint32_t)hcap;
            sa.thr2 = thr2;
            sa.part_count = c->part_count.as<uint32_t>();
            sa.part_score = c->part_score.as<double>();
            sa.tickets = blk_tot + 2 * (size_t)nblk + 2;
            const uint32_t slices = std::max<uint32_t>(1u, std::min<uint32_t>(1536u / chunks, (uint32_t)hcap));
            HIP_TRY(hipEventRecord(c->ev0, c->stream));
            HIP_TRY(launch_score(kind, sa, slices, c->stream));
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            FinalizeArgs fa;
            fa.num_hyp = sa.num_hyp;
            fa.hyp_capacity = (uint32_t)hcap;
            fa.chunks = chunks;
            fa.n_points = N;
            fa.thr2 = thr2;
            fa.part_count = sa.part_count;
            fa.part_score = sa.part_score;
            fa.count = c->count.as<uint32_t>();
            fa.score = c->score.as<double>();
            uint32_t *blk_max = c->blk_best.as<uint32_t>();
            double *blk_min = reinterpret_cast<double *>(c->blk_best.as<char>() + 1024);
            HIP_TRY(c->blk_best.ensure(1024 + sizeof(double) * 256));
            blk_max = c->blk_best.as<uint32_t>();
            blk_min = reinterpret_cast<double *>(c->blk_best.as<char>() + 1024);
            const uint32_t init_max = (uint32_t)std::min<uint64_t>(best_min_inl, 0xffffffffu);
            HIP_TRY(launch_finalize_records(fa, sa.slots, ga.models, blk_max, blk_min, init_max, best_min_score,
                                            c->rec_meta.as<RecordMeta>(), c->rec_models.as<double>(), kRecordCap, d_ctl,
                                            c->h_rec_meta.dev<RecordMeta>(), c->h_gather_out.dev<double>(), kRecordFirst,
                                            c->stream));
            // the listed candidates once more, summed like the reference sums them (decisions are taken on these)
            SeqScoreArgs qa;
            qa.pts = p->ps;
            qa.models = ga.models;
            qa.cand = c->rec_meta.as<RecordMeta>();
            qa.num = &d_ctl->num_records;
            qa.cap = kRecordCap;
            qa.thr2 = thr2;
            qa.count = nullptr;
            qa.score = nullptr;
            qa.host_count = nullptr;
            qa.host_score = nullptr;
            qa.host_cand = c->h_rec_meta.dev<RecordMeta>();
            qa.host_cap = kRecordFirst;
            qa.ctl_src = d_ctl; // the last kernel of the batch mirrors the control block into pinned host memory
            qa.ctl_host = c->h_small.dev<BatchCtl>();
            HIP_TRY(launch_score_seq(kind, qa, c->stream));
            b.ctl_mirrored = true;
        }
        b.pos_after = pos_after;
        b.device_positions = device_positions;
        b.hcap = hcap;
        b.d_ctl = d_ctl;
        b.prosac_at_batch_start = prosac_at_batch_start;
        return PL_OK;
    }

    // The candidates (sorted by hypothesis index, scores summed in the reference's order) through the rule of
    // ransac_impl.h:113-123: those that improve the running best become `imps`; the last one of an iteration seeds
    // the local optimisation.
    void keep_improving(const RecordMeta *meta, const uint32_t *order_, uint32_t n, uint32_t first_iteration) {
        for (uint32_t a = 0; a < n; ++a) {
            const RecordMeta &m = meta[order_[a]];
            const bool more = m.count > best_min_inl, better = m.score < best_min_score;
            if (!(more || better))
                continue;
            if (more)
                best_min_inl = m.count;
            if (better)
                best_min_score = m.score;
            Improving im;
            im.iter = first_iteration + m.slot / (uint32_t)MAXM;
            im.slot = m.slot;
            im.count = m.count;
            im.score = m.score;
            im.lo_seed = false;
            im.gather = order_[a];
            if (!imps.empty() && imps.back().iter != im.iter)
                imps.back().lo_seed = true;
            imps.push_back(im);
        }
        if (!imps.empty())
            imps.back().lo_seed = true;
    }

    // ---- the one synchronisation of the batch; retry decisions; improving hypotheses of this rank's share ----
    int collect_improving(Batch &b) {
        const uint32_t B = b.B, lo_g = b.lo_g, hi_g = b.hi_g, Bl = b.Bl;
        (void)B, (void)lo_g, (void)hi_g, (void)Bl;
        const size_t hcap = b.hcap;
        BatchCtl *const d_ctl = b.d_ctl;
        uint64_t &pos_after = b.pos_after;
        const bool device_positions = b.device_positions;
        const ProsacSampler &prosac_at_batch_start = b.prosac_at_batch_start;
        (void)hcap;
        BatchCtl *h_ctl = c->h_small.as<BatchCtl>();
        RecordMeta *h_meta = c->h_rec_meta.as<RecordMeta>(); // the first kRecordFirst candidates: written by k_score_seq
        double *h_recm = c->h_gather_out.as<double>();
        if (!b.ctl_mirrored)
            HIP_TRY(hipMemcpyAsync(h_ctl, d_ctl, sizeof(BatchCtl), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(wait_stream(c));
        if (device_positions) {
            if (h_ctl->orbit_error) { // evaluated window too small / too many redraws: redo this batch
                force_host_positions = true;
                return kRedoBatch;
            }
            pos_after = h_ctl->pos_after;
            static const bool check_positions = std::getenv("POSELIB_AMD_CHECK_POSITIONS") != nullptr;
            if (check_positions) { // diagnostic: the device's orbit walk against the sequential walk on the host
                std::vector<uint32_t> dev(B), host(B);
                HIP_TRY(hipMemcpy(dev.data(), c->positions.p, sizeof(uint32_t) * B, hipMemcpyDeviceToHost));
                const uint64_t after = sample_positions_k(K, ro.seed, pos, N, B, host.data());
                for (uint32_t i = 0; i < B; ++i)
                    if (dev[i] != host[i]) {
                        std::fprintf(stderr, "poselib_amd: sampler position mismatch: N %u K %d batch %u at draw %llu, iteration %u: device %u host %u\n",
                                     N, K, B, (unsigned long long)pos, i, dev[i], host[i]);
                        return fail(PL_ERR_HIP, "device sampler positions differ from the sequential walk");
                    }
                if (after != pos_after) {
                    std::fprintf(stderr, "poselib_amd: sampler end position mismatch: device %llu host %llu\n",
                                 (unsigned long long)pos_after, (unsigned long long)after);
                    return fail(PL_ERR_HIP, "device sampler end position differs from the sequential walk");
                }
            }
        }
        const bool overflow = h_ctl->gen_overflow != 0;
        if (overflow && !sh) { // an iteration produced more models than the reserved slots: redo with 40
            MAXM = max_models(kind);
            prosac_sampler = prosac_at_batch_start;
            return kRedoBatch;
        } // (sharded: the ranks agree on the retry in the exchange below)
        force_host_positions = false;
        uint32_t H = h_ctl->num_hyp; // sharded: replaced by the sum over the ranks after the exchange
        const uint32_t H_local = H;
        if (Bl > 0 && !overflow) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            st->score_kernel_ms += ms;
            st->score_kernel_launches++;
            st->nan_hypotheses += h_ctl->nan_hyp;
        }
        const uint64_t b0_inl = best_min_inl; // state of the sequential loop at the start of the batch
        const double b0_score = best_min_score;

        // ---- pass 1 result: the improving hypotheses, in (iteration, model) order ----
        imps.clear();
        const double *h_rec = h_recm;
        const uint32_t nrec = overflow ? 0u : h_ctl->num_records;
        if (overflow) {
            // nothing to report: every rank redoes the batch with more slots per iteration
        } else if (!host_records && nrec <= kRecordCap) {
            if (nrec > kRecordFirst) {
                HIP_TRY(hipMemcpyAsync(h_meta, c->rec_meta.p, sizeof(RecordMeta) * nrec, hipMemcpyDeviceToHost,
                                       c->stream));
                HIP_TRY(hipMemcpyAsync(h_recm, c->rec_models.p, sizeof(double) * kModelStride * nrec,
                                       hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(wait_stream(c));
            }
            order.resize(nrec);
            for (uint32_t a = 0; a < nrec; ++a)
                order[a] = a;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return h_meta[x].k < h_meta[y].k; });
            keep_improving(h_meta, order.data(), nrec, (uint32_t)(it + lo_g));
        } else {
            // fallback (candidate list overflow, or POSELIB_AMD_HOST_RECORDS=1): scan every (tree-order) score on the
            // host with the same margin as k_records, have the candidates re-scored in the reference's summation
            // order, then apply the exact rule like above
            HIP_TRY(c->h_num_models.ensure(sizeof(uint32_t) * (Bl + 1)));
            HIP_TRY(c->h_count.ensure(sizeof(uint32_t) * hcap));
            HIP_TRY(c->h_score.ensure(sizeof(double) * hcap));
            uint32_t *h_nm = c->h_num_models.as<uint32_t>();
            if (Bl)
                HIP_TRY(hipMemcpyAsync(h_nm, c->num_models.p, sizeof(uint32_t) * Bl, hipMemcpyDeviceToHost, c->stream));
            if (H) {
                HIP_TRY(hipMemcpyAsync(c->h_count.p, c->count.p, sizeof(uint32_t) * H, hipMemcpyDeviceToHost,
                                       c->stream));
                HIP_TRY(hipMemcpyAsync(c->h_score.p, c->score.p, sizeof(double) * H, hipMemcpyDeviceToHost,
                                       c->stream));
            }
            HIP_TRY(wait_stream(c));
            const uint32_t *h_cnt = c->h_count.as<uint32_t>();
            const double *h_sc = c->h_score.as<double>();
            std::vector<RecordMeta> cand;
            uint64_t run_inl = best_min_inl;
            double run_score = best_min_score;
            uint32_t k = 0;
            for (uint32_t i = 0; i < Bl; ++i)
                for (uint32_t m = 0; m < h_nm[i]; ++m, ++k) {
                    if (!(h_cnt[k] > run_inl || h_sc[k] < run_score * (1.0 + 1e-9)))
                        continue;
                    run_inl = std::max<uint64_t>(run_inl, h_cnt[k]);
                    run_score = std::min(run_score, h_sc[k]);
                    RecordMeta rm;
                    rm.k = k;
                    rm.slot = i * MAXM + m;
                    rm.count = h_cnt[k];
                    rm.pad = 0;
                    rm.score = h_sc[k];
                    cand.push_back(rm);
                }
            const uint32_t nc = (uint32_t)cand.size();
            HIP_TRY(c->h_rec_meta.ensure(sizeof(RecordMeta) * std::max<uint32_t>(nc, kRecordCap)));
            HIP_TRY(c->h_gather_out.ensure(sizeof(double) * kModelStride * std::max<uint32_t>(nc, kRecordCap)));
            h_meta = c->h_rec_meta.as<RecordMeta>();
            double *dst = c->h_gather_out.as<double>();
            if (nc) {
                HIP_TRY(c->rec_meta.ensure(sizeof(RecordMeta) * std::max<uint32_t>(nc, kRecordCap)));
                HIP_TRY(c->num_hyp.ensure(sizeof(uint32_t)));
                std::memcpy(h_meta, cand.data(), sizeof(RecordMeta) * nc);
                HIP_TRY(hipMemcpyAsync(c->rec_meta.p, h_meta, sizeof(RecordMeta) * nc, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipMemcpyAsync(c->num_hyp.p, &nc, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                SeqScoreArgs qa;
                qa.pts = p->ps;
                qa.models = c->models.as<double>();
                qa.cand = c->rec_meta.as<RecordMeta>();
                qa.num = c->num_hyp.as<uint32_t>();
                qa.cap = nc;
                qa.thr2 = thr2;
                qa.count = nullptr;
                qa.score = nullptr;
                qa.host_count = nullptr;
                qa.host_score = nullptr;
                qa.host_cand = nullptr;
                qa.host_cap = 0;
                HIP_TRY(launch_score_seq(kind, qa, c->stream));
                HIP_TRY(hipMemcpyAsync(h_meta, c->rec_meta.p, sizeof(RecordMeta) * nc, hipMemcpyDeviceToHost, c->stream));
                for (uint32_t a = 0; a < nc; ++a)
                    HIP_TRY(hipMemcpyAsync(dst + (size_t)a * kModelStride,
                                           c->models.as<double>() + (size_t)cand[a].slot * kModelStride,
                                           sizeof(double) * kModelStride, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(wait_stream(c));
            }
            order.resize(nc);
            for (uint32_t a = 0; a < nc; ++a)
                order[a] = a;
            keep_improving(h_meta, order.data(), nc, (uint32_t)(it + lo_g));
            h_rec = dst;
        }

        b.overflow = overflow;
        b.H = H;
        b.H_local = H_local;
        b.b0_inl = b0_inl;
        b.b0_score = b0_score;
        b.h_rec = h_rec;
        return PL_OK;
    }

    // ---- sharded runs: first exchange step of the batch (all-gather of the ranks' improving hypotheses) ----
    int exchange_improving(Batch &b) {
        const bool overflow = b.overflow;
        const uint32_t H_local = b.H_local;
        const uint64_t b0_inl = b.b0_inl;
        const double b0_score = b.b0_score;
        const ProsacSampler &prosac_at_batch_start = b.prosac_at_batch_start;
        uint32_t &H = b.H;
        const double *&h_rec = b.h_rec;
        // ---- sharded: the first exchange step of the batch.  Every rank contributes the improving hypotheses of its
        // range (improving w.r.t. the batch-start state and its own earlier hypotheses - a superset of what the
        // sequential loop keeps); the merged list is filtered with the true running best, rank by rank, i.e. in
        // iteration order.  From here on every rank holds the same list and does the same thing. ----
        if (sh) {
            const uint32_t nloc = (uint32_t)imps.size();
            auto pack = [&](unsigned char *dst, uint32_t first, uint32_t count) {
                for (uint32_t a = 0; a < count; ++a) {
                    WireRec w;
                    std::memset(&w, 0, sizeof(w));
                    if (first + a < nloc) {
                        const Improving &im = imps[first + a];
                        w.iter_off = (uint32_t)(im.iter - it);
                        w.count = im.count;
                        w.score = im.score;
                        std::memcpy(w.model, h_rec + (size_t)im.gather * kModelStride, sizeof(w.model));
                    }
                    std::memcpy(dst + (size_t)a * sizeof(WireRec), &w, sizeof(w));
                }
            };
            const size_t bytes1 = sizeof(WireHead) + (size_t)kWireFirst * sizeof(WireRec);
            wire_send.assign(bytes1, 0);
            wire_recv.assign(bytes1 * G, 0);
            WireHead head{nloc, overflow ? 1u : 0u, H_local, 0u};
            std::memcpy(wire_send.data(), &head, sizeof(head));
            pack(wire_send.data() + sizeof(WireHead), 0, kWireFirst);
            if (sh->allgather(sh->user, wire_send.data(), wire_recv.data(), bytes1) != 0)
                return fail(PL_ERR_COMM, "all-gather callback failed");
            std::vector<WireHead> heads(G);
            bool any_overflow = false;
            uint32_t max_n = 0;
            uint64_t H_sum = 0;
            for (uint32_t r = 0; r < G; ++r) {
                std::memcpy(&heads[r], wire_recv.data() + (size_t)r * bytes1, sizeof(WireHead));
                any_overflow = any_overflow || heads[r].gen_overflow != 0;
                max_n = std::max(max_n, heads[r].n);
                H_sum += heads[r].H;
            }
            if (any_overflow) { // some rank ran out of model slots: all ranks redo the batch with 40 per iteration
                best_min_inl = b0_inl;
                best_min_score = b0_score;
                MAXM = max_models(kind);
                prosac_sampler = prosac_at_batch_start;
                return kRedoBatch;
            }
            const unsigned char *recs = wire_recv.data() + sizeof(WireHead);
            size_t rank_stride = bytes1;
            std::vector<unsigned char> recv2;
            if (max_n > kWireFirst) { // rare: a second message with every rank's full list
                const size_t bytes2 = (size_t)max_n * sizeof(WireRec);
                std::vector<unsigned char> send2(bytes2, 0);
                recv2.assign(bytes2 * G, 0);
                pack(send2.data(), 0, max_n);
                if (sh->allgather(sh->user, send2.data(), recv2.data(), bytes2) != 0)
                    return fail(PL_ERR_COMM, "all-gather callback failed");
                recs = recv2.data();
                rank_stride = bytes2;
            }
            imps.clear();
            merged_models.clear();
            uint64_t run_inl = b0_inl;
            double run_score = b0_score;
            for (uint32_t r = 0; r < G; ++r)
                for (uint32_t a = 0; a < heads[r].n; ++a) {
                    WireRec w;
                    std::memcpy(&w, recs + (size_t)r * rank_stride + (size_t)a * sizeof(WireRec), sizeof(w));
                    const bool more = w.count > run_inl, better = w.score < run_score; // ransac_impl.h:114-116
                    if (!(more || better))
                        continue;
                    if (more)
                        run_inl = w.count;
                    if (better)
                        run_score = w.score;
                    Improving im;
                    im.iter = (uint32_t)(it + w.iter_off);
                    im.slot = 0;
                    im.count = w.count;
                    im.score = w.score;
                    im.lo_seed = false;
                    im.gather = (uint32_t)imps.size();
                    if (!imps.empty() && imps.back().iter != im.iter)
                        imps.back().lo_seed = true;
                    imps.push_back(im);
                    merged_models.insert(merged_models.end(), w.model, w.model + kModelStride);
                }
            if (!imps.empty())
                imps.back().lo_seed = true;
            best_min_inl = run_inl;
            best_min_score = run_score;
            h_rec = merged_models.data();
            H = (uint32_t)std::min<uint64_t>(H_sum, 0xffffffffu);
        }
        return PL_OK;
    }

    // Sharded runs: the local optimisations of a batch are dealt out to the ranks (job j to rank j mod world; the
    // kernels are deterministic, so it does not matter which device refines a seed) and their results - refined record,
    // inlier count, score - are all-gathered: the second (and last) exchange step of a batch, ~0.2 KB per job.
    int run_refinements_sharded() {
        struct WireJob {
            double record_out[kModelStride];
            double score;
            uint32_t count, skipped;
        };
        const uint32_t nj = (uint32_t)jobs.size();
        const uint32_t per_rank = (nj + G - 1) / G;
        std::vector<RefineJob> mine;
        for (uint32_t j = grank; j < nj; j += G)
            mine.push_back(jobs[j]);
        if (!mine.empty()) {
            const int rc = run_refinements(c, p, mine, true, thr2);
            if (rc != PL_OK)
                return rc;
        }
        const size_t bytes = (size_t)per_rank * sizeof(WireJob);
        wire_send.assign(bytes, 0);
        wire_recv.assign(bytes * G, 0);
        for (uint32_t a = 0; a < (uint32_t)mine.size(); ++a) {
            WireJob w;
            std::memset(&w, 0, sizeof(w));
            std::memcpy(w.record_out, mine[a].record_out, sizeof(w.record_out));
            w.score = mine[a].score;
            w.count = mine[a].count;
            w.skipped = mine[a].skipped ? 1u : 0u;
            std::memcpy(wire_send.data() + (size_t)a * sizeof(WireJob), &w, sizeof(w));
        }
        if (sh->allgather(sh->user, wire_send.data(), wire_recv.data(), bytes) != 0)
            return fail(PL_ERR_COMM, "all-gather callback failed");
        for (uint32_t j = 0; j < nj; ++j) {
            WireJob w;
            std::memcpy(&w, wire_recv.data() + (size_t)(j % G) * bytes + (size_t)(j / G) * sizeof(WireJob), sizeof(w));
            std::memcpy(jobs[j].record_out, w.record_out, sizeof(w.record_out));
            jobs[j].score = w.score;
            jobs[j].count = w.count;
            jobs[j].skipped = w.skipped != 0;
        }
        return PL_OK;
    }

    // ---- the local optimisations the batch's improving hypotheses trigger (ransac_impl.h:127-131: the last improving
    // hypothesis of an iteration seeds one) ----
    void make_jobs(const Batch &b) {
        const uint32_t ni = (uint32_t)imps.size();
        jobs.clear();
        for (uint32_t a = 0; a < ni; ++a)
            if (imps[a].lo_seed) {
                imps[a].job = (int)jobs.size();
                jobs.push_back(make_lo_job(b.h_rec + (size_t)imps[a].gather * kModelStride));
            }
    }

    // ---- batched local optimisations of the batch, then the replay of the sequential loop over it ----
    int refine_and_replay(Batch &b) {
        // ---- device: every triggered LO of the batch as one batched launch, then re-scored ----
        make_jobs(b);
        if (!jobs.empty()) {
            const int rc = sh ? run_refinements_sharded() : run_refinements(c, p, jobs, true, thr2);
            if (rc != PL_OK)
                return rc;
        }
        return replay(b, nullptr);
    }

    // ---- host pass 2: replay the sequential loop over this batch (ransac_impl.h:180-188) with the refined and
    // re-scored models in `jobs`.  host_offsets: pinned mirror of the per-iteration hypothesis offsets of the batch (group
    // launches write one); nullptr: the one entry that is needed is fetched from the device ----
    // (defer_offset: group launches - the caller fetches offsets[deferred_offset] of all stopping problems in one go and
    // books it into st->hypotheses; mirroring the whole table to pinned memory cost 0.4 MB of PCIe writes per problem and batch)
    bool defer_offset = false;
    int64_t deferred_offset = -1;
    int replay(Batch &b, const uint32_t *host_offsets) {
        const uint32_t B = b.B, lo_g = b.lo_g, hi_g = b.hi_g, Bl = b.Bl;
        (void)B, (void)lo_g, (void)hi_g, (void)Bl;
        const uint32_t H = b.H, H_local = b.H_local;
        const double *const h_rec = b.h_rec;
        const uint64_t pos_after = b.pos_after;
        st->iterations_evaluated += B;
        const uint32_t ni = (uint32_t)imps.size();

        // The stop rule can only change at LO events, so the replay hops from event to event.
        uint64_t cursor = it;          // next iteration whose stop check has not been made yet
        uint64_t stop_at = it + B;     // first iteration NOT replayed
        for (uint32_t a = 0; a < ni; ++a) {
            const Improving &im = imps[a];
            const uint64_t first_stop = std::max<uint64_t>(std::max<uint64_t>(ro.min_iterations, dyn_max) + 1, cursor);
            if (first_stop <= im.iter) {
                stopped = true;
                stop_at = first_stop;
                break;
            }
            if (im.score < st->model_score) { // :126-131
                st->model_score = im.score;
                st->num_inliers = im.count;
                std::memcpy(best_record, h_rec + (size_t)im.gather * kModelStride, sizeof(double) * kModelStride);
                best_from_lo = false; // (a minimal model: its bits are the reference's)
            }
            if (im.lo_seed) {
                const int rc_lo = after_lo(jobs[im.job]);
                if (rc_lo != PL_OK)
                    return rc_lo;
            }
            cursor = (uint64_t)im.iter + 1;
        }
        if (!stopped) {
            const uint64_t first_stop = std::max<uint64_t>(std::max<uint64_t>(ro.min_iterations, dyn_max) + 1, cursor);
            if (first_stop < it + B) {
                stopped = true;
                stop_at = first_stop;
            }
        }
        if (stop_at < it + B) { // hypotheses of the replayed iterations only
            uint32_t upto = 0;
            if (stop_at >= it + hi_g) {
                upto = H_local;
            } else if (stop_at > it + lo_g) {
                if (host_offsets) {
                    upto = host_offsets[stop_at - it - lo_g];
                } else if (defer_offset && !sh) {
                    deferred_offset = (int64_t)(stop_at - it - lo_g);
                } else {
                    HIP_TRY(hipMemcpyAsync(&upto, c->offsets.as<uint32_t>() + (stop_at - it - lo_g), sizeof(uint32_t),
                                           hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(wait_stream(c));
                }
            }
            uint64_t total = upto;
            if (sh) { // sum of the ranks' shares (8 bytes each; once per run)
                std::vector<uint64_t> all(G, 0);
                if (sh->allgather(sh->user, &total, all.data(), sizeof(uint64_t)) != 0)
                    return fail(PL_ERR_COMM, "all-gather callback failed");
                total = 0;
                for (uint32_t r = 0; r < G; ++r)
                    total += all[r];
            }
            st->hypotheses += total;
        } else {
            st->hypotheses += H;
        }
        it = stop_at;
        pos = pos_after;
        return PL_OK;
    }

    // ---- size of the next batch (false: the loop is over) ----
    uint64_t grow = 0;
    void begin_loop() {
        // batch capacity: bounded by the scratch the model records need
        grow = std::max<uint64_t>(ro.min_iterations + 2, 512);
        grow = (grow + 63) / 64 * 64;
        if (prosac)
            prosac_sampler.init(ro.seed, N, K, ro.max_prosac_iterations);
    }
    // iterations the loop is known to need at least from here (plan_batch's bound, without its side effects)
    uint64_t needed_now() const {
        uint64_t needed = std::max<uint64_t>(ro.min_iterations, dyn_max) + 1;
        needed = std::min<uint64_t>(needed, ro.max_iterations);
        return (needed > it) ? needed - it : 1;
    }
    bool plan_batch(Batch &b) {
        if (stopped || it >= ro.max_iterations)
            return false;
        if (it > ro.min_iterations && it > dyn_max) { // stop rule at the top of the next iteration (:182)
            stopped = true;
            return false;
        }
        // the loop cannot stop before max(min_iterations, dynamic_max_iter) + 1 iterations
        uint64_t needed = std::max<uint64_t>(ro.min_iterations, dyn_max) + 1;
        needed = std::min<uint64_t>(needed, ro.max_iterations);
        needed = (needed > it) ? needed - it : 1;
        const uint32_t cap = std::min<uint32_t>(131072u, 1048576u / (uint32_t)MAXM); // scratch size (<= 200 MB of records)
        b = Batch();
        b.B = (uint32_t)std::min<uint64_t>({needed, (uint64_t)cap, grow, ro.max_iterations - it});
        grow = std::min<uint64_t>(grow * 2, 131072u);
        // this rank's share of the batch (the whole batch on a single device)
        b.lo_g = (uint32_t)((uint64_t)b.B * grank / G);
        b.hi_g = (uint32_t)((uint64_t)b.B * (grank + 1) / G);
        b.Bl = b.hi_g - b.lo_g;
        return true;
    }

    int run() {
        std::memset(st, 0, sizeof(*st));
        st->model_score = std::numeric_limits<double>::max();
        const double t_start = now_s();
        bool mask_done = false;
        if (N >= (uint32_t)K) { // ransac_impl.h:161-163
            st->num_inliers = 0;
            if (ro.score_initial_model) {
                const int rc = score_initial_model();
                if (rc != PL_OK)
                    return rc;
            }
            begin_loop();
            Batch b;
            while (plan_batch(b)) {
                int rc = enqueue_batch(b);
                if (rc == PL_OK)
                    rc = collect_improving(b);
                if (rc == PL_OK && sh)
                    rc = exchange_improving(b);
                if (rc == kRedoBatch)
                    continue;
                if (rc == PL_OK)
                    rc = refine_and_replay(b);
                if (rc != PL_OK)
                    return rc;
            }
            st->iterations = it;

            // ---- final refinement of the best model (ransac_impl.h:190-198; model_score is not updated), chained on the
            // device with the choice refined / incumbent and the inlier mask of the returned model (ransac.cc:55, 152,
            // 259, 311): one synchronisation ----
            {
                std::vector<RefineJob> fin{make_lo_job(best_record)};
                MaskTail tail;
                tail.incumbent_score = st->model_score;
                tail.incumbent_rec = best_record;
                tail.thr2 = thr2;
                tail.host_mask = inliers;
                int rc = run_refinements(c, p, fin, true, thr2, &tail);
                if (rc != PL_OK)
                    return rc;
                st->refinements++;
                if (fin[0].score < st->model_score) {
                    std::memcpy(best_record, fin[0].record_out, sizeof(double) * kModelStride);
                    st->num_inliers = fin[0].count;
                }
                mask_done = true;
            }
        }

        // ---- inlier mask of the returned model when the loop did not run (too few points: ransac_impl.h:161-163) ----
        if (N > 0 && !mask_done) {
            HIP_TRY(c->mask.ensure(N));
            HIP_TRY(c->tmp_model.ensure(sizeof(double) * kModelStride));
            HIP_TRY(hipMemcpyAsync(c->tmp_model.p, best_record, sizeof(double) * kModelStride, hipMemcpyHostToDevice,
                                   c->stream));
            HIP_TRY(c->h_mask.ensure(N));
            HIP_TRY(launch_mask(kind, p->ps, c->tmp_model.as<double>(), thr2, c->mask.as<uint8_t>(),
                                inliers ? c->h_mask.dev<uint8_t>() : nullptr, c->stream));
            HIP_TRY(wait_stream(c));
            if (inliers)
                std::memcpy(inliers, c->h_mask.p, N);
        }
        st->seconds = now_s() - t_start;
        return PL_OK;
    }
};

int ransac_core(Context *c, const pl_problem *p, const pl_robust_options *o, double *best_record /* in/out */,
                uint8_t *inliers, pl_ransac_stats *st) {
    RansacRun run(c, p, o, best_record, inliers, st);
    return run.run();
}

int validate_options(const pl_robust_options *o, bool focal_entry = false) {
    if (!o)
        return fail(PL_ERR_INVALID, "options pointer is null");
    if (o->min_fov != o->min_fov) // (a record shorter than this library's - built against an older header - reads garbage here)
        return fail(PL_ERR_INVALID, "min_fov is NaN (options record not filled by pl_default_robust_options, or built against another PL_ABI_VERSION?)");
    // estimate_focal_length: pl_estimate_absolute_pose only (robust.cc:47-54 -> ransac_pnpf, driver_focal.inc)
    if (focal_entry && o->estimate_focal_length && !o->tangent_sampson && !o->estimate_extra_params)
        return PL_OK;
    // (bundle.refine_*: used by the absolute-pose front-end's final bundle, robust.cc:103-123; the other front-ends'
    // refiners have no camera to move, the reference ignores the flags there and so does this library)
    if (o->tangent_sampson || o->estimate_focal_length || o->estimate_extra_params)
        return fail(PL_ERR_UNSUPPORTED, "tangent-Sampson errors, estimate_extra_params and - outside pl_estimate_absolute_pose - "
                                        "estimate_focal_length are outside the accelerated hot path");
    return PL_OK;
}

int make_problem(Context *c, int kind, const double *a, const double *b, size_t n, pl_problem *p) {
    if (kind < 0 || kind > 3)
        return fail(PL_ERR_INVALID, "unknown problem kind");
    if (n > 0x7fffffffu)
        return fail(PL_ERR_INVALID, "too many correspondences");
    p->kind = kind;
    p->device = c->device;
    p->n = (uint32_t)n;
    p->d_pts = nullptr;
    const int nd = point_doubles(kind);
    const int da = 2, db = (kind == EST_ABS) ? 3 : 2;
    std::memset(&p->ps, 0, sizeof(p->ps));
    p->ps.n = (uint32_t)n;
    if (n == 0)
        return PL_OK;
    std::vector<double> soa((size_t)nd * n);
    double amax = 0;
    for (size_t i = 0; i < n; ++i) {
        for (int d = 0; d < da; ++d) {
            soa[(size_t)d * n + i] = a[da * i + d];
            const double v = std::fabs(a[da * i + d]);
            amax = (v > amax || v != v) ? v : amax; // NaN propagates and disables the pre-filter
        }
        for (int d = 0; d < db; ++d) {
            soa[(size_t)(da + d) * n + i] = b[db * i + d];
            if (kind != EST_ABS) { // two-view: the bound covers all four coordinates (pl_prefilter.h, fp16 Sampson form)
                const double v = std::fabs(b[db * i + d]);
                amax = (v > amax || v != v) ? v : amax;
            }
        }
    }
    HIP_TRY(hipMalloc((void **)&p->d_pts, sizeof(double) * nd * n));
    hipError_t up = hipMemcpyAsync(p->d_pts, soa.data(), sizeof(double) * nd * n, hipMemcpyHostToDevice, c->stream);
    if (up == hipSuccess)
        up = hipStreamSynchronize(c->stream);
    if (up != hipSuccess) {
        (void)hipFree(p->d_pts);
        p->d_pts = nullptr;
        return fail(PL_ERR_HIP, "upload of the correspondences", up);
    }
    for (int d = 0; d < nd; ++d)
        p->ps.a[d] = p->d_pts + (size_t)d * n;
    p->ps.xy_absmax = std::nextafter((float)amax, std::numeric_limits<float>::infinity());
    return PL_OK;
}
void free_problem(pl_problem *p) {
    if (p->d_pts && !p->borrowed)
        (void)hipFree(p->d_pts);
    p->d_pts = nullptr;
}

// Upper bound of the largest |coordinate| the prepared two-view points will have (pl_prefilter.h, fp16 Sampson form), from
// the raw points on the host: linear cameras / the normalisation (x - c) / scale round at most once per operation.
// +inf when it cannot be told without the device (OPENCV un-projection) or a coordinate is NaN.
float host_two_view_absmax(const PrepareArgs &pa, const double *a, const double *b, size_t n) {
    const float inf = std::numeric_limits<float>::infinity();
    double m = 0.0;
    auto take = [&](double v) { m = (v > m || v != v) ? v : m; };
    if (pa.mode == 1) {
        const CameraParams *cams[2] = {&pa.cam1, &pa.cam2};
        const double *pts[2] = {a, b};
        for (int k = 0; k < 2; ++k) {
            const CameraParams &cam = *cams[k];
            double cx = 0, cy = 0, fx = 1, fy = 1;
            if (cam.model_id == CAM_SIMPLE_PINHOLE)
                fx = fy = cam.p[0], cx = cam.p[1], cy = cam.p[2];
            else if (cam.model_id == CAM_PINHOLE)
                fx = cam.p[0], fy = cam.p[1], cx = cam.p[2], cy = cam.p[3];
            else if (cam.model_id != CAM_NULL)
                return inf;
            for (size_t i = 0; i < n; ++i) {
                take(std::fabs((pts[k][2 * i] - cx) / fx));
                take(std::fabs((pts[k][2 * i + 1] - cy) / fy));
            }
        }
    } else if (pa.mode == 2) {
        const double c1x = pa.centred ? pa.c1x : 0.0, c1y = pa.centred ? pa.c1y : 0.0;
        const double c2x = pa.centred ? pa.c2x : 0.0, c2y = pa.centred ? pa.c2y : 0.0;
        for (size_t i = 0; i < n; ++i) {
            take(std::fabs((a[2 * i] - c1x) / pa.scale));
            take(std::fabs((a[2 * i + 1] - c1y) / pa.scale));
            take(std::fabs((b[2 * i] - c2x) / pa.scale));
            take(std::fabs((b[2 * i + 1] - c2y) / pa.scale));
        }
    } else {
        return inf;
    }
    if (m != m)
        return inf;
    m = m * (1.0 + 1e-12);
    return std::nextafter((float)m, inf);
}

// One-shot front-ends: the user's AoS buffers go to the device as they are and k_prepare (pipeline.hip) writes the SoA
// block into the context's arena - per-point un-projection / normalisation on the GPU, no
// hipMalloc / hipFree per call.  The problem is valid until the next make_problem_prepared on this thread.
// resident: the raw correspondences of this call are already in c->raw_a / raw_b (second preparation of the same
// inputs: no upload); lm_only: the problem is only refined on, never scored - no max|x| read-back, no synchronisation
int make_problem_prepared(Context *c, int kind, const double *a, const double *b, size_t n, const PrepareArgs &pa,
                          pl_problem *p, bool resident = false, bool lm_only = false) {
    if (kind < 0 || kind > 3)
        return fail(PL_ERR_INVALID, "unknown problem kind");
    if (n > 0x7fffffffu)
        return fail(PL_ERR_INVALID, "too many correspondences");
    p->kind = kind;
    p->device = c->device;
    p->n = (uint32_t)n;
    p->d_pts = nullptr;
    p->borrowed = true;
    std::memset(&p->ps, 0, sizeof(p->ps));
    p->ps.n = (uint32_t)n;
    if (n == 0)
        return PL_OK;
    const int nd = point_doubles(kind);
    const int db = (kind == EST_ABS) ? 3 : 2;
    HIP_TRY(c->pts_arena.ensure(sizeof(double) * nd * n));
    HIP_TRY(c->absmax.ensure(sizeof(unsigned long long)));
    HIP_TRY(c->h_absmax.ensure(sizeof(unsigned long long)));
    // max|x| of the prepared points is read back only for absolute-pose problems of non-linear cameras (below): no fill dispatch
    // in front of the others (k_prepare's atomic max then lands on a word nobody reads)
    const bool reads_absmax = !(lm_only || kind != EST_ABS) && !(pa.mode == 0 && pa.cam1.model_id != CAM_OPENCV);
    if (reads_absmax)
        HIP_TRY(hipMemsetAsync(c->absmax.p, 0, sizeof(unsigned long long), c->stream));
    if (!resident) {
        const size_t bytes_a = sizeof(double) * 2 * n, bytes_b = sizeof(double) * db * n;
        if (bytes_a + bytes_b <= ((size_t)8 << 20)) {
            // through a pinned, mapped block k_prepare reads directly: no copy dispatches, and no copy from pageable memory (which pins
            // and unpins the caller's buffer under the process's memory-map lock - what front-end calls from many host threads
            // queued on, DESIGN 4 "focal estimators")
            HIP_TRY(c->h_in.ensure(bytes_a + bytes_b));
            std::memcpy(c->h_in.p, a, bytes_a);
            std::memcpy(c->h_in.as<char>() + bytes_a, b, bytes_b);
            c->raw_src_a = c->h_in.dev<double>();
            c->raw_src_b = reinterpret_cast<const double *>(c->h_in.dev<char>() + bytes_a);
        } else {
            HIP_TRY(c->raw_a.ensure(bytes_a));
            HIP_TRY(c->raw_b.ensure(bytes_b));
            HIP_TRY(hipMemcpyAsync(c->raw_a.p, a, bytes_a, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->raw_b.p, b, bytes_b, hipMemcpyHostToDevice, c->stream));
            c->raw_src_a = c->raw_a.as<double>();
            c->raw_src_b = c->raw_b.as<double>();
        }
    }
    if (!c->raw_src_a || !c->raw_src_b)
        return fail(PL_ERR_INVALID, "second preparation without a first");
    HIP_TRY(launch_prepare(c->raw_src_a, c->raw_src_b, (uint32_t)n, pa, c->pts_arena.as<double>(),
                           c->absmax.as<unsigned long long>(), c->stream));
    p->d_pts = c->pts_arena.as<double>();
    for (int d = 0; d < nd; ++d)
        p->ps.a[d] = p->d_pts + (size_t)d * n;
    if (lm_only || kind != EST_ABS) { // (stream order puts the next kernels behind k_prepare)
        // two-view: the coordinate bound that admits the matrix-core form of the Sampson filter (large problems only: the
        // O(N) host pass is not worth it below the size that form starts at)
        p->ps.xy_absmax = std::numeric_limits<float>::infinity();
        if (!lm_only && n >= 1024 && a && b) // (relative pose, fundamental matrix, homography)
            p->ps.xy_absmax = host_two_view_absmax(pa, a, b, n);
        return PL_OK;
    }
    if (pa.mode == 0 && pa.cam1.model_id != CAM_OPENCV) {
        // linear cameras: an upper bound of max(|x|, |y|) after un-projection from the raw pixels on the host - no
        // read-back, no synchronisation (the un-projected coordinate is (px - c) / f up to one rounding)
        const CameraParams &cam = pa.cam1;
        double cx = 0, cy = 0, fx = 1, fy = 1;
        if (cam.model_id == CAM_SIMPLE_PINHOLE)
            fx = fy = cam.p[0], cx = cam.p[1], cy = cam.p[2];
        else if (cam.model_id == CAM_PINHOLE)
            fx = cam.p[0], fy = cam.p[1], cx = cam.p[2], cy = cam.p[3];
        double m = 0.0;
        for (size_t i = 0; i < n; ++i) {
            const double u = std::fabs((a[2 * i] - cx) / fx), v = std::fabs((a[2 * i + 1] - cy) / fy);
            m = (u > m || u != u) ? u : m; // NaN propagates (and disables the pre-filter)
            m = (v > m || v != v) ? v : m;
        }
        m = m * (1.0 + 1e-12);
        p->ps.xy_absmax = std::nextafter((float)m, std::numeric_limits<float>::infinity());
        return PL_OK;
    }
    HIP_TRY(hipMemcpyAsync(c->h_absmax.p, c->absmax.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c));
    double amax;
    std::memcpy(&amax, c->h_absmax.p, sizeof(double));
    p->ps.xy_absmax = std::nextafter((float)amax, std::numeric_limits<float>::infinity());
    return PL_OK;
}
PrepareArgs prepare_unproject(const CameraParams &c1, const CameraParams *c2) {
    PrepareArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.mode = c2 ? 1 : 0;
    pa.cam1 = c1;
    if (c2)
        pa.cam2 = *c2;
    pa.scale = 1.0;
    return pa;
}

// user model <-> record
void record_from_pose(const pl_camera_pose *pose, bool essential, double *rec) {
    Quat q;
    q.w = pose->q[0], q.x = pose->q[1], q.y = pose->q[2], q.z = pose->q[3];
    store_pose_model_q(rec, q, v3(pose->t[0], pose->t[1], pose->t[2]), essential);
}
void pose_from_record(const double *rec, pl_camera_pose *pose) {
    for (int i = 0; i < 4; ++i)
        pose->q[i] = rec[i];
    for (int i = 0; i < 3; ++i)
        pose->t[i] = rec[4 + i];
}
Mat3 mat_from_colmajor(const double *m) {
    Mat3 A;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            A.m[3 * i + j] = m[3 * j + i];
    return A;
}
void mat_to_colmajor(const Mat3 &A, double *m) {
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            m[3 * j + i] = A.m[3 * i + j];
}
Mat3 transpose3(const Mat3 &A) {
    Mat3 T;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T.m[3 * i + j] = A.m[3 * j + i];
    return T;
}
void normalize_frobenius(Mat3 &A) {
    double s = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            s += A.m[3 * i + j] * A.m[3 * i + j];
    const double n = std::sqrt(s);
    for (int i = 0; i < 9; ++i)
        A.m[i] /= n;
}

int run_with_model(Context *c, pl_problem *p, const pl_robust_options *o, void *model, uint8_t *inliers,
                   pl_ransac_stats *st, double *record_out = nullptr) {
    double rec[kModelStride];
    const bool pose_kind = (p->kind == EST_ABS || p->kind == EST_REL);
    if (o->ransac.score_initial_model) {
        if (pose_kind)
            record_from_pose(static_cast<const pl_camera_pose *>(model), p->kind == EST_REL, rec);
        else
            store_matrix_model(rec, mat_from_colmajor(static_cast<const double *>(model)));
    } else {
        identity_record(p->kind, rec);
    }
    int rc = ransac_core(c, p, o, rec, inliers, st);
    if (rc != PL_OK)
        return rc;
    if (pose_kind) {
        pose_from_record(rec, static_cast<pl_camera_pose *>(model));
    } else {
        Mat3 M;
        for (int i = 0; i < 9; ++i)
            M.m[i] = rec[kMatOff + i];
        mat_to_colmajor(M, static_cast<double *>(model));
    }
    if (record_out)
        std::memcpy(record_out, rec, sizeof(rec));
    return PL_OK;
}

// Final polish on the inliers (device mask from ransac_core is still in c->mask).
int final_refine(Context *c, pl_problem *p, const double *record_in, const LMOptions &opt, const CameraParams &cam,
                 double point_scale, double *record_out, double *params_out, int cam_flags = 0,
                 CameraParams *cam_out = nullptr) {
    RefineJob j;
    std::memcpy(j.record_in, record_in, sizeof(j.record_in));
    j.opt = opt;
    j.cam = cam;
    j.point_scale = point_scale;
    j.prefilter_thr2 = 0.0;
    j.d_mask = c->mask.as<uint8_t>();
    j.cam_flags = cam_flags;
    std::vector<RefineJob> jobs{j};
    int rc = run_refinements(c, p, jobs, false, 0.0);
    if (rc != PL_OK)
        return rc;
    if (cam_out)
        *cam_out = jobs[0].cam_out;
    std::memcpy(record_out, jobs[0].record_out, sizeof(double) * kModelStride);
    if (params_out)
        std::memcpy(params_out, jobs[0].params_out, sizeof(double) * kParamDoubles);
    return PL_OK;
}

// PoseLib/robust/utils.cc:584-644 with normalize_scale = shared_scale = true.  Only the two order-dependent reductions
// (centroids, mean distance) are summed here, sequentially like the reference; the per-point part - subtract the
// centroid, divide by the scale - is applied by k_prepare on the device with the same two operations per coordinate.
double normalization_of(const double *x1, const double *x2, size_t n, bool centroid, Mat3 &T1, Mat3 &T2,
                        PrepareArgs &pa) {
    for (int i = 0; i < 9; ++i)
        T1.m[i] = T2.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
    double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
    if (centroid) {
        for (size_t k = 0; k < n; ++k) {
            c1x += x1[2 * k], c1y += x1[2 * k + 1];
            c2x += x2[2 * k], c2y += x2[2 * k + 1];
        }
        c1x /= static_cast<double>(n), c1y /= static_cast<double>(n);
        c2x /= static_cast<double>(n), c2y /= static_cast<double>(n);
        T1.m[2] = -c1x, T1.m[5] = -c1y;
        T2.m[2] = -c2x, T2.m[5] = -c2y;
    }
    double scale = 0.0;
    for (size_t k = 0; k < n; ++k) {
        double a0 = x1[2 * k], a1 = x1[2 * k + 1], b0 = x2[2 * k], b1 = x2[2 * k + 1];
        if (centroid)
            a0 -= c1x, a1 -= c1y, b0 -= c2x, b1 -= c2y;
        scale += std::sqrt(a0 * a0 + a1 * a1);
        scale += std::sqrt(b0 * b0 + b1 * b1);
    }
    scale /= std::sqrt(2) * n;
    const double f = 1.0 / scale;
    for (int i = 0; i < 6; ++i) {
        T1.m[i] *= f;
        T2.m[i] *= f;
    }
    std::memset(&pa, 0, sizeof(pa));
    pa.mode = 2;
    pa.centred = centroid ? 1 : 0;
    pa.c1x = c1x, pa.c1y = c1y, pa.c2x = c2x, pa.c2y = c2y;
    pa.scale = scale;
    return scale;
}

#include "driver_focal.inc"
#include "driver_sfocal.inc"
#include "driver_group.inc"
#include "driver_focal_group.inc"

} // namespace

// =============================================================================================== C-ABI
extern "C" {

const char *pl_version(void) { return "poselib_amd 0.1 (gfx950)"; }
int pl_abi_version(void) { return PL_ABI_VERSION; }
const char *pl_last_error(void) { return g_err.c_str(); }

void pl_default_ransac_options(pl_ransac_options *o) {
    o->max_iterations = 100000;
    o->min_iterations = 1000;
    o->dyn_num_trials_mult = 3.0;
    o->success_prob = 0.9999;
    o->seed = 0;
    o->progressive_sampling = 0;
    o->score_initial_model = 0;
    o->max_prosac_iterations = 100000;
}
void pl_default_bundle_options(pl_bundle_options *o) {
    std::memset(o, 0, sizeof(*o));
    o->max_iterations = 100;
    o->loss_type = LOSS_CAUCHY;
    o->lambda_update = 0;
    o->damping = 0;
    o->loss_scale = 1.0;
    o->gradient_tol = 1e-12;
    o->step_tol = 1e-8;
    o->relative_cost_tol = 1e-10;
    o->initial_lambda = 1e-3;
    o->min_lambda = 1e-10;
    o->max_lambda = 1e10;
    o->lambda_factor = 10.0;
}
void pl_default_robust_options(pl_robust_options *o, int kind) {
    std::memset(o, 0, sizeof(*o));
    pl_default_ransac_options(&o->ransac);
    pl_default_bundle_options(&o->bundle);
    o->max_error = (kind == 0) ? 12.0 : 1.0;
    o->min_fov = 5.0; // types.h:126
}

int pl_set_lm_mode(int mode) {
    const int prev = pl::get_lm_mode();
    pl::set_lm_mode(mode);
    return prev;
}

int pl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}
int pl_set_device(int device) {
    g_requested_device = device;
    Context *c;
    return get_context(&c);
}

int pl_problem_create(int kind, const double *a, const double *b, size_t n, pl_problem **out) {
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    pl_problem *p = new pl_problem();
    rc = make_problem(c, kind, a, b, n, p);
    if (rc != PL_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return PL_OK;
}
void pl_problem_destroy(pl_problem *p) {
    if (!p)
        return;
    free_problem(p);
    delete p;
}
int pl_ransac_run(pl_problem *p, const pl_robust_options *opt, void *model, uint8_t *inliers, pl_ransac_stats *stats) {
    if (!p || !model)
        return fail(PL_ERR_INVALID, "problem / model pointer is null");
    int rc = validate_options(opt);
    if (rc != PL_OK)
        return rc;
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (p->device != c->device)
        return fail(PL_ERR_INVALID, "problem lives on another device than the calling thread's");
    pl_ransac_stats local;
    return run_with_model(c, p, opt, model, inliers, stats ? stats : &local);
}

int pl_ransac_run_sharded(pl_problem *p, const pl_robust_options *opt, const pl_shard *shard, void *model,
                          uint8_t *inliers, pl_ransac_stats *stats) {
    if (!shard || shard->world < 1 || shard->rank < 0 || shard->rank >= shard->world)
        return fail(PL_ERR_INVALID, "shard: need 0 <= rank < world");
    if (shard->world > 1 && !shard->allgather)
        return fail(PL_ERR_INVALID, "shard: all-gather callback missing");
    g_shard = shard;
    const int rc = pl_ransac_run(p, opt, model, inliers, stats);
    g_shard = nullptr;
    return rc;
}

int pl_score_model(pl_problem *p, const void *model, double max_error, uint64_t *inlier_count, double *score) {
    if (!p || !model)
        return fail(PL_ERR_INVALID, "problem / model pointer is null");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (p->device != c->device)
        return fail(PL_ERR_INVALID, "problem lives on another device than the calling thread's");
    double rec[kModelStride];
    if (p->kind == EST_ABS || p->kind == EST_REL)
        record_from_pose(static_cast<const pl_camera_pose *>(model), p->kind == EST_REL, rec);
    else
        store_matrix_model(rec, mat_from_colmajor(static_cast<const double *>(model)));
    HIP_TRY(c->tmp_model.ensure(sizeof(double) * kModelStride));
    HIP_TRY(hipMemcpyAsync(c->tmp_model.p, rec, sizeof(rec), hipMemcpyHostToDevice, c->stream));
    rc = enqueue_score_records(c, p, c->tmp_model.as<double>(), 1, max_error * max_error, false);
    if (rc != PL_OK)
        return rc;
    HIP_TRY(wait_stream(c));
    if (inlier_count)
        *inlier_count = c->h_count.as<uint32_t>()[0];
    if (score)
        *score = c->h_score.as<double>()[0];
    return PL_OK;
}

// Diagnostic: an arbitrary list of models through the STREAMING scorer of the main loop (k_gather_models -> k_shadow16 ->
// k_score_mfma / k_score_queue -> k_finalize2), i.e. through the conservative pre-filters, instead of the sequential
// scorer pl_score_model uses.  Lets tests feed adversarial models / points to the filters on the device.
int pl_debug_score_stream(pl_problem *p, const void *models, size_t n, double max_error, uint32_t *counts,
                          double *scores, int32_t *path_used) {
    if (!p || (!models && n))
        return fail(PL_ERR_INVALID, "problem / models pointer is null");
    if (n > (1u << 22))
        return fail(PL_ERR_INVALID, "too many models");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (p->device != c->device)
        return fail(PL_ERR_INVALID, "problem lives on another device than the calling thread's");
    if (path_used)
        *path_used = 0;
    if (n == 0)
        return PL_OK;
    const uint32_t H = (uint32_t)n;
    const bool pose_kind = (p->kind == EST_ABS || p->kind == EST_REL);
    std::vector<double> recs((size_t)H * kModelStride);
    std::vector<uint32_t> ident(H);
    for (uint32_t k = 0; k < H; ++k) {
        ident[k] = k;
        if (pose_kind)
            record_from_pose(static_cast<const pl_camera_pose *>(models) + k, p->kind == EST_REL, &recs[(size_t)k * kModelStride]);
        else
            store_matrix_model(&recs[(size_t)k * kModelStride], mat_from_colmajor(static_cast<const double *>(models) + 9 * (size_t)k));
    }
    const double thr2 = max_error * max_error;
    ScoreArgs sa;
    set_prefilter(sa, p, thr2);
    const bool on_mfma = score_uses_mfma(p->kind, p->n, sa.pf);
    const uint32_t chunks = score_chunks(p->kind, p->n, true, on_mfma);
    HIP_TRY(c->models.ensure(sizeof(double) * kModelStride * H));
    HIP_TRY(c->slots.ensure(sizeof(uint32_t) * H));
    HIP_TRY(c->shadow.ensure(sizeof(float) * 16 * H));
    HIP_TRY(c->compact64.ensure(sizeof(double) * kModelDoubles * H));
    HIP_TRY(c->ctl.ensure(sizeof(BatchCtl) + 64 + sizeof(uint32_t) * chunks));
    HIP_TRY(c->part_count.ensure(sizeof(uint32_t) * chunks * H));
    HIP_TRY(c->part_score.ensure(sizeof(double) * chunks * H));
    HIP_TRY(c->count.ensure(sizeof(uint32_t) * H));
    HIP_TRY(c->score.ensure(sizeof(double) * H));
    BatchCtl hc;
    std::memset(&hc, 0, sizeof(hc));
    hc.num_hyp = H;
    BatchCtl *d_ctl = c->ctl.as<BatchCtl>();
    HIP_TRY(hipMemsetAsync(d_ctl, 0, sizeof(BatchCtl) + 64 + sizeof(uint32_t) * chunks, c->stream));
    HIP_TRY(hipMemcpyAsync(d_ctl, &hc, sizeof(hc), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->models.p, recs.data(), sizeof(double) * recs.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->slots.p, ident.data(), sizeof(uint32_t) * H, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_gather_models(d_ctl, c->slots.as<uint32_t>(), c->models.as<double>(), H, c->shadow.as<float>(),
                                 c->compact64.as<double>(), c->stream));
    sa.pts = p->ps;
    sa.models = c->models.as<double>();
    sa.slots = c->slots.as<uint32_t>();
    sa.shadow16 = nullptr;
    int path = sa.pf.enabled ? 1 : 0;
    if (on_mfma && p->kind == EST_ABS) {
        HIP_TRY(c->shadow16.ensure(((size_t)H + kAbs16Pad) * kAbs16Bytes));
        HIP_TRY(launch_shadow16(&d_ctl->num_hyp, c->shadow.as<float>(), H, sa.pf.g16, sa.pf.c16, sa.pf.thr,
                                c->shadow16.p, c->stream));
        sa.shadow16 = c->shadow16.p;
        path = 2;
    } else if (on_mfma && p->kind == EST_HOM) {
        HIP_TRY(c->shadow16.ensure(((size_t)H + kHom16Pad) * kHom16Bytes));
        HIP_TRY(launch_hom16(d_ctl, c->slots.as<uint32_t>(), c->models.as<double>(), H, sa.pf.h16, c->shadow16.p, c->stream));
        sa.shadow16 = c->shadow16.p;
        path = 2;
    } else if (on_mfma) {
        HIP_TRY(c->shadow16.ensure(((size_t)H + kSampson16Pad) * kSampson16Bytes));
        HIP_TRY(launch_sampson16(d_ctl, c->slots.as<uint32_t>(), c->models.as<double>(), H, c->shadow16.p, c->stream));
        sa.shadow16 = c->shadow16.p;
        path = 2;
    }
    sa.shadow = c->shadow.as<float>();
    sa.compact64 = c->compact64.as<double>();
    sa.num_hyp = &d_ctl->num_hyp;
    sa.hyp_capacity = H;
    sa.thr2 = thr2;
    sa.part_count = c->part_count.as<uint32_t>();
    sa.part_score = c->part_score.as<double>();
    sa.tickets = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(d_ctl) + sizeof(BatchCtl) + 64);
    const uint32_t slices = std::max<uint32_t>(1u, std::min<uint32_t>(1536u / chunks, H));
    HIP_TRY(launch_score(p->kind, sa, slices, c->stream));
    FinalizeArgs fa;
    fa.num_hyp = sa.num_hyp;
    fa.hyp_capacity = H;
    fa.chunks = chunks;
    fa.n_points = p->n;
    fa.thr2 = thr2;
    fa.part_count = sa.part_count;
    fa.part_score = sa.part_score;
    fa.count = c->count.as<uint32_t>();
    fa.score = c->score.as<double>();
    HIP_TRY(launch_finalize(fa, c->stream));
    if (counts)
        HIP_TRY(hipMemcpyAsync(counts, c->count.p, sizeof(uint32_t) * H, hipMemcpyDeviceToHost, c->stream));
    if (scores)
        HIP_TRY(hipMemcpyAsync(scores, c->score.p, sizeof(double) * H, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c));
    if (path_used)
        *path_used = path;
    return PL_OK;
}

int pl_debug_device_math(int fn, const double *x, size_t n, double *out) {
    if ((!x || !out) && n)
        return fail(PL_ERR_INVALID, "null pointer");
    if (n > (1u << 28))
        return fail(PL_ERR_INVALID, "too many values");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (n == 0)
        return PL_OK;
    struct Scratch { // freed on every path
        void *p = nullptr;
        ~Scratch() {
            if (p)
                (void)hipFree(p);
        }
    } din, dout;
    HIP_TRY(hipMalloc(&din.p, sizeof(double) * n));
    HIP_TRY(hipMalloc(&dout.p, sizeof(double) * n));
    HIP_TRY(hipMemcpyAsync(din.p, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_device_math(fn, static_cast<const double *>(din.p), (uint32_t)n, static_cast<double *>(dout.p), c->stream));
    HIP_TRY(hipMemcpyAsync(out, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PL_OK;
}

int pl_refine_model(pl_problem *p, const pl_bundle_options *opt, const pl_camera *camera, const uint8_t *mask,
                    void *model, uint32_t *lm_iterations) {
    if (!p || !model || !opt)
        return fail(PL_ERR_INVALID, "problem / options / model pointer is null");
    if (p->kind == EST_ABS && camera && active_cam_flags(to_cam(camera).model_id, *opt))
        return fail(PL_ERR_INVALID, "refine_* moves the camera: call pl_bundle_adjust_camera (camera in / out)");
    if (camera && !camera_supported(camera))
        return fail(PL_ERR_UNSUPPORTED, "camera model not supported (NULL, SIMPLE_PINHOLE, PINHOLE, OPENCV)");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (p->device != c->device)
        return fail(PL_ERR_INVALID, "problem lives on another device than the calling thread's");
    RefineJob j;
    const bool pose_kind = (p->kind == EST_ABS || p->kind == EST_REL);
    if (pose_kind)
        record_from_pose(static_cast<const pl_camera_pose *>(model), p->kind == EST_REL, j.record_in);
    else
        store_matrix_model(j.record_in, mat_from_colmajor(static_cast<const double *>(model)));
    j.opt = to_lm(*opt);
    j.cam = to_cam(camera);
    j.point_scale = 1.0;
    j.prefilter_thr2 = 0.0;
    j.d_mask = nullptr;
    if (mask && p->n) {
        HIP_TRY(c->mask.ensure(p->n));
        HIP_TRY(hipMemcpyAsync(c->mask.p, mask, p->n, hipMemcpyHostToDevice, c->stream));
        j.d_mask = c->mask.as<uint8_t>();
    }
    std::vector<RefineJob> jobs{j};
    rc = run_refinements(c, p, jobs, false, 0.0);
    if (rc != PL_OK)
        return rc;
    if (lm_iterations)
        *lm_iterations = c->h_tasks.as<LMTask>()[0].iterations;
    if (pose_kind) {
        pose_from_record(jobs[0].record_out, static_cast<pl_camera_pose *>(model));
    } else {
        Mat3 M;
        for (int i = 0; i < 9; ++i)
            M.m[i] = jobs[0].record_out[kMatOff + i];
        mat_to_colmajor(M, static_cast<double *>(model));
    }
    return PL_OK;
}

// bundle_adjust(x, X, Image *image, BundleOptions) (the Image overload, robust/bundle.cc:94-113) on an absolute-pose
// problem holding PIXEL coordinates: the pose and - per opt->refine_focal_length / refine_principal_point /
// refine_extra_params - the camera's intrinsics, both in / out.
int pl_bundle_adjust_camera(pl_problem *p, const pl_bundle_options *opt, pl_camera *camera, const uint8_t *mask,
                            pl_camera_pose *pose, uint32_t *lm_iterations) {
    if (!p || !pose || !opt || !camera)
        return fail(PL_ERR_INVALID, "problem / options / camera / pose pointer is null");
    if (p->kind != EST_ABS)
        return fail(PL_ERR_INVALID, "pl_bundle_adjust_camera refines absolute poses");
    if (!camera_supported(camera))
        return fail(PL_ERR_UNSUPPORTED, "camera model not supported (NULL, SIMPLE_PINHOLE, PINHOLE, OPENCV)");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (p->device != c->device)
        return fail(PL_ERR_INVALID, "problem lives on another device than the calling thread's");
    RefineJob j;
    record_from_pose(pose, false, j.record_in);
    j.opt = to_lm(*opt);
    j.cam = to_cam(camera);
    j.cam_flags = active_cam_flags(j.cam.model_id, *opt);
    j.point_scale = 1.0;
    j.prefilter_thr2 = 0.0;
    j.d_mask = nullptr;
    if (mask && p->n) {
        HIP_TRY(c->mask.ensure(p->n));
        HIP_TRY(hipMemcpyAsync(c->mask.p, mask, p->n, hipMemcpyHostToDevice, c->stream));
        j.d_mask = c->mask.as<uint8_t>();
    }
    std::vector<RefineJob> jobs{j};
    rc = run_refinements(c, p, jobs, false, 0.0);
    if (rc != PL_OK)
        return rc;
    if (lm_iterations)
        *lm_iterations = c->h_tasks.as<LMTask>()[0].iterations;
    pose_from_record(jobs[0].record_out, pose);
    if (j.cam_flags)
        for (int i = 0; i < camera->num_params && i < 12; ++i)
            camera->params[i] = jobs[0].cam_out.p[i];
    return PL_OK;
}

static int ransac_oneshot(int kind, const double *a, const double *b, size_t n, const pl_robust_options *opt,
                          void *model, uint8_t *inliers, pl_ransac_stats *stats) {
    int rc = validate_options(opt);
    if (rc != PL_OK)
        return rc;
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    pl_problem p;
    rc = make_problem(c, kind, a, b, n, &p);
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    rc = run_with_model(c, &p, opt, model, inliers, stats ? stats : &local);
    free_problem(&p);
    return rc;
}
int pl_ransac_pnp(const double *x, const double *X, size_t n, const pl_robust_options *opt, pl_camera_pose *pose,
                  uint8_t *inliers, pl_ransac_stats *stats) {
    return ransac_oneshot(EST_ABS, x, X, n, opt, pose, inliers, stats);
}
int pl_ransac_pnpf(const double *x, const double *X, size_t n, const pl_robust_options *opt, pl_camera_pose *pose, double *focal,
                   uint8_t *inliers, pl_ransac_stats *stats) {
    if (!opt || !pose || !focal)
        return fail(PL_ERR_INVALID, "null argument");
    pl_robust_options o = *opt;
    o.estimate_focal_length = 1;
    int rc = validate_options(&o, /*focal_entry=*/true);
    if (rc != PL_OK)
        return rc;
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    pl_problem p;
    rc = make_problem(c, EST_ABS, x, X, n, &p);
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    rc = run_focal(c, &p, &o, pose, focal, inliers, stats ? stats : &local, x, nullptr);
    free_problem(&p);
    return rc;
}
int pl_ransac_shared_focal_relpose(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, pl_camera_pose *pose,
                                   double *focal, uint8_t *inliers, pl_ransac_stats *stats) {
    if (!opt || !pose || !focal)
        return fail(PL_ERR_INVALID, "null argument");
    int rc = validate_options(opt);
    if (rc != PL_OK)
        return rc;
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    pl_problem p;
    rc = make_problem(c, EST_FUND, x1, x2, n, &p); // (two-view layout: x1, y1, x2, y2)
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    rc = run_shared_focal(c, &p, opt, pose, focal, inliers, stats ? stats : &local);
    free_problem(&p);
    return rc;
}
int pl_refine_shared_focal_relpose(const double *x1, const double *x2, size_t n, const pl_bundle_options *opt, pl_camera_pose *pose,
                                   double *focal, uint32_t *lm_iterations) {
    if (!opt || !pose || !focal)
        return fail(PL_ERR_INVALID, "null argument");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    pl_problem p;
    rc = make_problem(c, EST_FUND, x1, x2, n, &p);
    if (rc != PL_OK)
        return rc;
    rc = refine_shared_focal(c, &p, to_lm(*opt), nullptr, pose, focal, lm_iterations);
    free_problem(&p);
    return rc;
}
int pl_ransac_relpose(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, pl_camera_pose *pose,
                      uint8_t *inliers, pl_ransac_stats *stats) {
    return ransac_oneshot(EST_REL, x1, x2, n, opt, pose, inliers, stats);
}
int pl_ransac_fundamental(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, double *F,
                          uint8_t *inliers, pl_ransac_stats *stats) {
    return ransac_oneshot(EST_FUND, x1, x2, n, opt, F, inliers, stats);
}
int pl_ransac_homography(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, double *H,
                         uint8_t *inliers, pl_ransac_stats *stats) {
    return ransac_oneshot(EST_HOM, x1, x2, n, opt, H, inliers, stats);
}

// ---------------------------------------------------------------------------- front-ends (robust.cc)
int pl_estimate_absolute_pose(const double *points2D, const double *points3D, size_t n, const pl_robust_options *opt,
                              pl_camera *camera, pl_camera_pose *pose, uint8_t *inliers, pl_ransac_stats *stats) {
    int rc = validate_options(opt, /*focal_entry=*/true);
    if (rc != PL_OK)
        return rc;
    if (!camera || !camera_supported(camera))
        return fail(PL_ERR_UNSUPPORTED, "camera model not supported (NULL, SIMPLE_PINHOLE, PINHOLE, OPENCV)");
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    // robust.cc:40-46 : un-project, rescale the threshold by 1/focal
    CameraParams cam = to_cam(camera);
    pl_robust_options scaled = *opt;
    double scale = 1.0 / camera_focal(camera);
    scaled.max_error *= scale;

    pl_problem p;
    rc = make_problem_prepared(c, EST_ABS, points2D, points3D, n, prepare_unproject(cam, nullptr), &p);
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    pl_ransac_stats *st = stats ? stats : &local;
    double rec[kModelStride];
    pl_bundle_options bundle = opt->bundle;
    if (opt->estimate_focal_length) { // robust.cc:47-54: ransac_pnpf on the un-projected points, the camera takes its focal length
        double focal = 1.0;
        rc = run_focal(c, &p, &scaled, pose, &focal, inliers, st, points2D, &cam);
        free_problem(&p);
        if (rc != PL_OK)
            return rc;
        camera_set_focal(camera, focal / scale);
        cam = to_cam(camera);
        bundle.refine_focal_length = 1; // "force refinement of focal in this case"
        record_from_pose(pose, false, rec);
    } else {
        rc = run_with_model(c, &p, &scaled, pose, inliers, st, rec);
        free_problem(&p);
        if (rc != PL_OK)
            return rc;
    }

    if (st->num_inliers > 3) { // robust.cc:103-123 : bundle over the inliers in focal-normalised pixels
        pl_problem pp;
        CameraParams raw_cam;
        std::memset(&raw_cam, 0, sizeof(raw_cam));
        raw_cam.model_id = CAM_NULL; // pixels as they are
        rc = make_problem_prepared(c, EST_ABS, points2D, points3D, n, prepare_unproject(raw_cam, nullptr), &pp,
                                   /*resident=*/true, /*lm_only=*/true);
        if (rc != PL_OK)
            return rc;
        scale = 1.0 / camera_focal(camera);
        pl_bundle_options b = bundle;
        b.loss_scale = opt->bundle.loss_scale * scale;
        CameraParams cs = cam;
        camera_rescale(cs, scale);
        double out[kModelStride];
        // bundle.refine_*: the intrinsics the model has among them move with the pose (bundle.cc:93-118)
        const int cam_flags = active_cam_flags(cs.model_id, bundle);
        CameraParams refined = cs;
        rc = final_refine(c, &pp, rec, to_lm(b), cs, scale, out, nullptr, cam_flags, &refined);
        free_problem(&pp);
        if (rc != PL_OK)
            return rc;
        pose_from_record(out, pose);
        // camera.rescale(scale) ... rescale(1/scale) round trip of the reference (robust.cc:119-121)
        CameraParams back = cam_flags ? refined : cs;
        camera_rescale(back, 1.0 / scale);
        for (int i = 0; i < camera->num_params && i < 12; ++i)
            camera->params[i] = back.p[i];
    }
    return PL_OK;
}

int pl_estimate_relative_pose(const double *x1, const double *x2, size_t n, const pl_camera *camera1,
                              const pl_camera *camera2, const pl_robust_options *opt, pl_camera_pose *pose,
                              uint8_t *inliers, pl_ransac_stats *stats) {
    int rc = validate_options(opt);
    if (rc != PL_OK)
        return rc;
    if (!camera1 || !camera2 || !camera_supported(camera1) || !camera_supported(camera2))
        return fail(PL_ERR_UNSUPPORTED, "camera model not supported (NULL, SIMPLE_PINHOLE, PINHOLE, OPENCV)");
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    // robust.cc:249-253, 286-292
    const double scale = 0.5 * (1.0 / camera_focal(camera1) + 1.0 / camera_focal(camera2));
    pl_robust_options scaled = *opt;
    scaled.max_error *= scale;
    scaled.bundle.loss_scale *= scale;
    const CameraParams c1 = to_cam(camera1), c2 = to_cam(camera2);
    pl_problem p;
    rc = make_problem_prepared(c, EST_REL, x1, x2, n, prepare_unproject(c1, &c2), &p);
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    pl_ransac_stats *st = stats ? stats : &local;
    double rec[kModelStride];
    rc = run_with_model(c, &p, &scaled, pose, inliers, st, rec);
    if (rc == PL_OK && st->num_inliers > 5) { // robust.cc:296-311
        CameraParams nc;
        std::memset(&nc, 0, sizeof(nc));
        nc.model_id = CAM_NULL;
        double out[kModelStride];
        rc = final_refine(c, &p, rec, to_lm(scaled.bundle), nc, 1.0, out, nullptr);
        if (rc == PL_OK)
            pose_from_record(out, pose);
    }
    free_problem(&p);
    return rc;
}

int pl_estimate_shared_focal_relative_pose(const double *x1, const double *x2, size_t n, const double *pp, const pl_robust_options *opt,
                                           pl_camera_pose *pose, double *focal, uint8_t *inliers, pl_ransac_stats *stats) {
    if (!pp || !pose || !focal)
        return fail(PL_ERR_INVALID, "null argument");
    int rc = validate_options(opt);
    if (rc != PL_OK)
        return rc;
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    pl_ransac_stats *st = stats ? stats : &local;
    // robust.cc:373-390
    PrepareArgs prep;
    const double scale = normalization_about(x1, x2, n, pp[0], pp[1], prep);
    pl_robust_options scaled = *opt;
    scaled.max_error /= scale;
    scaled.bundle.loss_scale /= scale;
    double f = *focal;
    if (opt->ransac.score_initial_model) // :392-397
        f = f / scale;
    pl_problem p;
    rc = make_problem_prepared(c, EST_FUND, x1, x2, n, prep, &p, false, /*lm_only=*/true);
    if (rc != PL_OK)
        return rc;
    rc = run_shared_focal(c, &p, &scaled, pose, &f, inliers, st);
    if (rc == PL_OK && st->num_inliers > 6) // :401-415: refine_shared_focal_relpose over the inliers (c->mask: the device mask)
        rc = refine_shared_focal(c, &p, to_lm(scaled.bundle), c->mask.as<uint8_t>(), pose, &f, nullptr);
    free_problem(&p);
    if (rc != PL_OK)
        return rc;
    *focal = f * scale; // :417 (the caller's cameras: SIMPLE_PINHOLE {focal, pp[0], pp[1]}, :418-421)
    return PL_OK;
}

int pl_estimate_fundamental(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, double *F,
                            uint8_t *inliers, pl_ransac_stats *stats) {
    int rc = validate_options(opt);
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    pl_ransac_stats *st = stats ? stats : &local;
    if (n < 7) { // robust.cc:548-550
        std::memset(st, 0, sizeof(*st));
        st->model_score = std::numeric_limits<double>::max();
        return PL_OK;
    }
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    Mat3 T1, T2;
    PrepareArgs prep;
    const double scale = normalization_of(x1, x2, n, !opt->real_focal_check, T1, T2, prep);
    pl_robust_options scaled = *opt;
    scaled.max_error /= scale;
    scaled.bundle.loss_scale /= scale;
    Mat3 Fm = mat_from_colmajor(F);
    if (opt->ransac.score_initial_model) { // robust.cc:566-569
        Fm = mul(mul(inverse3(transpose3(T2)), Fm), inverse3(T1));
        normalize_frobenius(Fm);
    }
    double Fcm[9];
    mat_to_colmajor(Fm, Fcm);
    pl_problem p;
    rc = make_problem_prepared(c, EST_FUND, x1, x2, n, prep, &p);
    if (rc != PL_OK)
        return rc;
    double rec[kModelStride];
    rc = run_with_model(c, &p, &scaled, Fcm, inliers, st, rec);
    if (rc == PL_OK && st->num_inliers > 7) { // robust.cc:573-588
        CameraParams nc;
        std::memset(&nc, 0, sizeof(nc));
        nc.model_id = CAM_NULL;
        double out[kModelStride];
        rc = final_refine(c, &p, rec, to_lm(scaled.bundle), nc, 1.0, out, nullptr);
        if (rc == PL_OK)
            std::memcpy(rec, out, sizeof(rec));
    }
    free_problem(&p);
    if (rc != PL_OK)
        return rc;
    for (int i = 0; i < 9; ++i)
        Fm.m[i] = rec[kMatOff + i];
    Fm = mul(mul(transpose3(T2), Fm), T1); // robust.cc:590-591
    normalize_frobenius(Fm);
    mat_to_colmajor(Fm, F);
    return PL_OK;
}

int pl_estimate_homography(const double *x1, const double *x2, size_t n, const pl_robust_options *opt, double *H,
                           uint8_t *inliers, pl_ransac_stats *stats) {
    int rc = validate_options(opt);
    if (rc != PL_OK)
        return rc;
    pl_ransac_stats local;
    pl_ransac_stats *st = stats ? stats : &local;
    if (n < 4) { // robust.cc:716-718
        std::memset(st, 0, sizeof(*st));
        st->model_score = std::numeric_limits<double>::max();
        return PL_OK;
    }
    Context *c;
    rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    Mat3 T1, T2;
    PrepareArgs prep;
    const double scale = normalization_of(x1, x2, n, true, T1, T2, prep);
    pl_robust_options scaled = *opt;
    scaled.max_error /= scale;
    scaled.bundle.loss_scale /= scale;
    Mat3 Hm = mat_from_colmajor(H);
    if (opt->ransac.score_initial_model) { // robust.cc:729-732
        Hm = mul(mul(T2, Hm), inverse3(T1));
        normalize_frobenius(Hm);
    }
    double Hcm[9];
    mat_to_colmajor(Hm, Hcm);
    pl_problem p;
    rc = make_problem_prepared(c, EST_HOM, x1, x2, n, prep, &p);
    if (rc != PL_OK)
        return rc;
    double rec[kModelStride];
    rc = run_with_model(c, &p, &scaled, Hcm, inliers, st, rec);
    if (rc == PL_OK && st->num_inliers > 4) { // robust.cc:736-751
        CameraParams nc;
        std::memset(&nc, 0, sizeof(nc));
        nc.model_id = CAM_NULL;
        double out[kModelStride];
        rc = final_refine(c, &p, rec, to_lm(scaled.bundle), nc, 1.0, out, nullptr);
        if (rc == PL_OK)
            std::memcpy(rec, out, sizeof(rec));
    }
    free_problem(&p);
    if (rc != PL_OK)
        return rc;
    for (int i = 0; i < 9; ++i)
        Hm.m[i] = rec[kMatOff + i];
    Hm = mul(mul(inverse3(T2), Hm), T1); // robust.cc:753-754
    normalize_frobenius(Hm);
    mat_to_colmajor(Hm, H);
    return PL_OK;
}

// ---------------------------------------------------------------------------- un-distortion stage
int pl_undistort_points(const pl_camera *camera, const double *points2D, size_t n, double *out) {
    if (!camera || !camera_supported(camera) || camera->model_id == CAM_NULL)
        return fail(PL_ERR_UNSUPPORTED, "camera model not supported (SIMPLE_PINHOLE, PINHOLE, OPENCV)");
    if (n > 0x7fffffffu)
        return fail(PL_ERR_INVALID, "too many points");
    if (n && (!points2D || !out))
        return fail(PL_ERR_INVALID, "points pointer is null");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (n == 0)
        return PL_OK;
    const CameraParams cam = to_cam(camera);
    double fx, fy, cx, cy;
    if (camera->model_id == CAM_SIMPLE_PINHOLE)
        fx = fy = cam.p[0], cx = cam.p[1], cy = cam.p[2];
    else
        fx = cam.p[0], fy = cam.p[1], cx = cam.p[2], cy = cam.p[3];
    HIP_TRY(c->raw_a.ensure(sizeof(double) * 2 * n));
    HIP_TRY(c->raw_b.ensure(sizeof(double) * 2 * n));
    HIP_TRY(hipMemcpyAsync(c->raw_a.p, points2D, sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_undistort(c->raw_a.as<double>(), (uint32_t)n, cam, fx, fy, cx, cy, c->raw_b.as<double>(), c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->raw_b.p, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c));
    return PL_OK;
}

// ---------------------------------------------------------------------------- minimal solvers
int pl_solve_batch(int kind, const double *in, size_t count, double *out_models, uint32_t *out_counts) {
    if (kind < 0 || kind > 3)
        return fail(PL_ERR_INVALID, "unknown solver kind");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (count == 0)
        return PL_OK;
    const int K = sample_size(kind), MAXM = max_models(kind);
    const size_t in_bytes = sizeof(double) * 6 * K * count;
    const size_t out_bytes = sizeof(double) * kModelStride * MAXM * count;
    HIP_TRY(c->solve_in.ensure(in_bytes));
    HIP_TRY(c->solve_out.ensure(out_bytes));
    HIP_TRY(c->solve_cnt.ensure(sizeof(uint32_t) * count));
    HIP_TRY(hipMemcpyAsync(c->solve_in.p, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_solve_batch(kind, c->solve_in.as<double>(), (uint32_t)count, c->solve_out.as<double>(),
                               c->solve_cnt.as<uint32_t>(), c->stream));
    HIP_TRY(hipMemcpyAsync(out_models, c->solve_out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_counts, c->solve_cnt.p, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c));
    return PL_OK;
}

// the two focal-length solvers: kind 0 P3.5Pf (in: x 4 x 2, X 4 x 3 per problem; <= 10 models), kind 1 the 6-point shared-focal
// solver (in: x1 6 x 3, x2 6 x 3 unit bearings per problem; <= 60 models); out_models: 8 doubles (q, t, focal) per slot
int pl_solve_focal_batch(int kind, const double *in, size_t count, double *out_models, uint32_t *out_counts) {
    if (kind < 0 || kind > 1)
        return fail(PL_ERR_INVALID, "unknown focal solver kind");
    if ((!in || !out_models || !out_counts) && count)
        return fail(PL_ERR_INVALID, "null argument");
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    if (count == 0)
        return PL_OK;
    if (count > 0x7fffffffu / 64u)
        return fail(PL_ERR_INVALID, "too many problems");
    const size_t per_in = kind == 0 ? 20 : 36, slots = kind == 0 ? (size_t)kFocalMaxModels : (size_t)kSFocalMaxModels;
    const size_t in_bytes = sizeof(double) * per_in * count, out_bytes = sizeof(FocalModel) * slots * count;
    HIP_TRY(c->solve_in.ensure(in_bytes));
    HIP_TRY(c->solve_out.ensure(out_bytes));
    HIP_TRY(c->solve_cnt.ensure(sizeof(uint32_t) * count));
    HIP_TRY(hipMemcpyAsync(c->solve_in.p, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->solve_out.p, 0, out_bytes, c->stream));
    if (kind == 0) {
        const uint32_t per_pass = (uint32_t)std::min<size_t>(count, 8192);
        HIP_TRY(c->focal_stage.ensure(focal_stage_bytes(per_pass)));
        HIP_TRY(launch_focal_solve(c->solve_in.as<double>(), (uint32_t)count, c->solve_out.as<FocalModel>(), c->solve_cnt.as<uint32_t>(),
                                   c->focal_stage.as<double>(), per_pass, c->stream));
    }
    else {
        const uint32_t per_pass = (uint32_t)std::min<size_t>(count, 8192);
        HIP_TRY(c->focal_stage.ensure(sfocal_stage_bytes(per_pass)));
        HIP_TRY(launch_sfocal_solve(c->solve_in.as<double>(), (uint32_t)count, c->solve_out.as<FocalModel>(), c->solve_cnt.as<uint32_t>(),
                                    c->focal_stage.as<double>(), per_pass, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(out_models, c->solve_out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_counts, c->solve_cnt.p, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c));
    return PL_OK;
}
static int focal_solve_one(int kind, const double *a, size_t na, const double *b, size_t nb, pl_camera_pose *out, double *focals) {
    if (!a || !b || !out || !focals)
        return fail(PL_ERR_INVALID, "null argument");
    std::vector<double> in(na + nb);
    std::memcpy(in.data(), a, sizeof(double) * na);
    std::memcpy(in.data() + na, b, sizeof(double) * nb);
    const size_t slots = kind == 0 ? (size_t)kFocalMaxModels : (size_t)kSFocalMaxModels;
    std::vector<double> models(8 * slots);
    uint32_t n = 0;
    int rc = pl_solve_focal_batch(kind, in.data(), 1, models.data(), &n);
    if (rc != PL_OK)
        return rc;
    for (uint32_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k)
            out[i].q[k] = models[8 * i + k];
        for (int k = 0; k < 3; ++k)
            out[i].t[k] = models[8 * i + 4 + k];
        focals[i] = models[8 * i + 7];
    }
    return (int)n;
}
int pl_p35pf(const double *x, const double *X, pl_camera_pose *out, double *focals) { return focal_solve_one(0, x, 8, X, 12, out, focals); }
int pl_relpose_6pt_shared_focal(const double *x1, const double *x2, pl_camera_pose *out, double *focals) {
    return focal_solve_one(1, x1, 18, x2, 18, out, focals);
}

static int solve_one(int kind, const double *a, const double *b, double *records, uint32_t *cnt) {
    const int K = sample_size(kind);
    std::vector<double> in(6 * K);
    std::memcpy(in.data(), a, sizeof(double) * 3 * K);
    std::memcpy(in.data() + 3 * K, b, sizeof(double) * 3 * K);
    return pl_solve_batch(kind, in.data(), 1, records, cnt);
}
int pl_p3p(const double *x, const double *X, pl_camera_pose *out) {
    double rec[4 * kModelStride];
    uint32_t n = 0;
    int rc = solve_one(EST_ABS, x, X, rec, &n);
    if (rc != PL_OK)
        return rc;
    for (uint32_t i = 0; i < n; ++i)
        pose_from_record(rec + i * kModelStride, out + i);
    return (int)n;
}
int pl_relpose_5pt(const double *x1, const double *x2, pl_camera_pose *out) {
    std::vector<double> rec(40 * kModelStride);
    uint32_t n = 0;
    int rc = solve_one(EST_REL, x1, x2, rec.data(), &n);
    if (rc != PL_OK)
        return rc;
    for (uint32_t i = 0; i < n; ++i)
        pose_from_record(rec.data() + i * kModelStride, out + i);
    return (int)n;
}
int pl_essential_matrix_5pt(const double *x1, const double *x2, double *E) {
    // kind 4 (internal): essential matrices before the motion decomposition
    Context *c;
    int rc = get_context(&c);
    if (rc != PL_OK)
        return rc;
    std::vector<double> in(30), rec(10 * kModelStride);
    std::memcpy(in.data(), x1, sizeof(double) * 15);
    std::memcpy(in.data() + 15, x2, sizeof(double) * 15);
    uint32_t n = 0;
    HIP_TRY(c->solve_in.ensure(sizeof(double) * 30));
    HIP_TRY(c->solve_out.ensure(sizeof(double) * kModelStride * 10));
    HIP_TRY(c->solve_cnt.ensure(sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(c->solve_in.p, in.data(), sizeof(double) * 30, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_solve_batch(4, c->solve_in.as<double>(), 1, c->solve_out.as<double>(), c->solve_cnt.as<uint32_t>(),
                               c->stream));
    HIP_TRY(hipMemcpyAsync(rec.data(), c->solve_out.p, sizeof(double) * kModelStride * 10, hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipMemcpyAsync(&n, c->solve_cnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c));
    for (uint32_t i = 0; i < n; ++i) {
        Mat3 M;
        for (int k = 0; k < 9; ++k)
            M.m[k] = rec[i * kModelStride + kMatOff + k];
        mat_to_colmajor(M, E + 9 * i);
    }
    return (int)n;
}
int pl_relpose_7pt(const double *x1, const double *x2, double *F) {
    double rec[3 * kModelStride];
    uint32_t n = 0;
    int rc = solve_one(EST_FUND, x1, x2, rec, &n);
    if (rc != PL_OK)
        return rc;
    for (uint32_t i = 0; i < n; ++i) {
        Mat3 M;
        for (int k = 0; k < 9; ++k)
            M.m[k] = rec[i * kModelStride + kMatOff + k];
        mat_to_colmajor(M, F + 9 * i);
    }
    return (int)n;
}
int pl_homography_4pt(const double *x1, const double *x2, double *H) {
    double rec[kModelStride];
    uint32_t n = 0;
    int rc = solve_one(EST_HOM, x1, x2, rec, &n);
    if (rc != PL_OK)
        return rc;
    if (n) {
        Mat3 M;
        for (int k = 0; k < 9; ++k)
            M.m[k] = rec[kMatOff + k];
        mat_to_colmajor(M, H);
    }
    return (int)n;
}

#include "driver_batch.inc"

} // extern "C"
