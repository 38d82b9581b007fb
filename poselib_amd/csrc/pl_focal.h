// poselib_amd - absolute pose with unknown focal length: the pieces of ransac_pnpf (PoseLib/robust/ransac.cc:58-75 with
// FocalAbsolutePoseEstimator, robust/estimators/absolute_pose.{h:69-113, cc:71-177}) that are shared by the kernels (focal.hip),
// the host driver (driver_focal.inc) and the test-only host build (tests/hostmath):
//   * the model of the estimator - an Image with a SIMPLE_PINHOLE camera {f, 0, 0} - as 8 doubles (q, t, f),
//   * the per-correspondence residual of compute_msac_score(Image, ...) (robust/utils.cc:66-98) and of get_inliers(Image, ...)
//     (utils.cc:385-399), in the reference's association order,
//   * the sequential loop of ransac_impl.h:157-201 replayed over batches of iterations that a back end evaluates (focal_lo_ransac).
// The minimal solver is pl_solver_p35pf.h.  Defaults of the class that no public entry point of the reference changes:
// solver P3.5Pf, refine_minimal_sample = filter_minimal_sample = false, inlier_scoring = true.
#pragma once
#include "pl_math.h"
#include "pl_sampler.h"

#include <limits>
#include <vector>

namespace pl {

constexpr int kFocalSample = 4;
constexpr int kFocalMaxModels = 10;
struct FocalModel {
    double q[4], t[3], f;
};
static_assert(sizeof(FocalModel) == 64, "8 doubles");

// utils.cc:76-94 with SIMPLE_PINHOLE project (camera_models.cc: f x + cx, cx = cy = 0 here: + 0.0 changes no squared residual)
PL_HD bool focal_reproj_inlier(const double *R, const double *t, double f, double x, double y, double X, double Y, double Z,
                               double thr2, double &r2) {
    const double z0 = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    const double z1 = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    const double z2 = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double inv = 1.0 / z2;
    const double e0 = f * (z0 * inv) - x;
    const double e1 = f * (z1 * inv) - y;
    r2 = e0 * e0 + e1 * e1;
    return (z2 > 0.0) & (r2 < thr2);
}
// utils.cc:385-399: hnormalized() divides
PL_HD bool focal_reproj_mask(const double *R, const double *t, double f, double x, double y, double X, double Y, double Z,
                             double thr2) {
    const double z0 = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    const double z1 = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    const double z2 = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double e0 = f * (z0 / z2) - x;
    const double e1 = f * (z1 / z2) - y;
    const double r2 = e0 * e0 + e1 * e1;
    return (r2 < thr2) & (z2 > 0.0);
}
PL_HD void focal_rotation(const FocalModel &m, double *R) {
    Quat q;
    q.w = m.q[0], q.x = m.q[1], q.y = m.q[2], q.z = m.q[3];
    const Mat3 Rm = quat_to_rotmat(q);
    for (int i = 0; i < 9; ++i)
        R[i] = Rm.m[i];
}

// ---- host side ----
// absolute_pose.cc:159-177
inline double focal_max_focal_length(const double *x, const double *y, size_t n, double min_fov) {
    if (min_fov <= 0)
        return -1.0;
    double max_coord = 0.0;
    for (size_t i = 0; i < n; ++i) {
        max_coord = std::max(max_coord, std::fabs(x[i]));
        max_coord = std::max(max_coord, std::fabs(y[i]));
    }
    const double min_fov_radians = min_fov * M_PI / 180.0;
    return max_coord / std::tan(min_fov_radians / 2.0);
}
// score_model (absolute_pose.cc:128-143) from the sum of the inliers' squared residuals and their number: the MSAC score of
// utils.cc:95 plus the outliers once more (inlier_scoring), associated as the reference writes them
inline double focal_finish_score(double inlier_sum, uint64_t count, uint64_t n, double max_error, double focal, double max_focal) {
    if (focal < 0)
        return std::numeric_limits<double>::max();
    double score = inlier_sum;
    score += static_cast<double>(n - count) * (max_error * max_error);
    score += static_cast<double>(n - count) * max_error * max_error;
    if (max_focal > 0 && focal > max_focal)
        score = std::numeric_limits<double>::max();
    return score;
}

struct FocalLoopOptions {
    uint64_t max_iterations, min_iterations, seed;
    double dyn_num_trials_mult, success_prob;
    bool score_initial_model;
    double max_error, max_focal;
    bool progressive_sampling = false;      // PROSAC (sampling.cc:85-136): the samples are drawn on the host, one ProsacSampler per run
    uint64_t max_prosac_iterations = 100000;
};
struct FocalLoopStats {
    uint64_t refinements = 0, iterations = 0, num_inliers = 0, hypotheses = 0, iterations_evaluated = 0;
    double inlier_ratio = 0, model_score = std::numeric_limits<double>::max();
};

// ransac_impl.h:43-73
inline uint64_t focal_dynamic_max_iter(uint64_t inl, uint64_t N, uint64_t K, double log_fail, double mult, uint64_t min_it,
                                       uint64_t max_it) {
    double p = 1.0;
    if (inl < K || N < K) {
        p = 0.0;
    } else {
        for (uint64_t i = 0; i < K; ++i)
            p *= static_cast<double>(inl - i) / static_cast<double>(N - i);
    }
    if (p >= 0.9999)
        return min_it;
    if (p <= 0.0001)
        return max_it;
    const uint64_t n = static_cast<uint64_t>(std::ceil(log_fail / std::log(1.0 - p) * mult));
    return std::max(min_it, std::min(max_it, n));
}

// draws consumed before each of `count` iterations (relative to pos), for samples of K distinct indices; returns the position
// after the last one (sampling.cc:45-60: duplicates are redrawn)
template <int K> inline uint64_t focal_sample_positions(uint64_t seed, uint64_t pos, uint64_t N, uint32_t count, uint32_t *out) {
    const uint64_t base = pos;
    for (uint32_t b = 0; b < count; ++b) {
        out[b] = (uint32_t)(pos - base);
        uint32_t idx[K];
        pos += draw_sample<K>(seed, pos, N, idx);
    }
    return pos;
}

// The sequential LO-RANSAC loop (ransac_impl.h:106-201) over batches of iterations, as a STATE MACHINE: advance() runs the loop up to
// the next thing only a back end can evaluate and returns what that is; the caller evaluates it (for one problem on its stream -
// focal_lo_ransac_t below - or for the members of a group with ONE launch sequence, driver_focal_group.inc), leaves the results in
// the response fields and calls advance() again.  Requests:
//   kMinimal      generate + score the batch [pos, positions[0 .. B)) (explicit_samples: B x kSample indices drawn here - PROSAC)
//                 -> num_models [B]; models, counts, sums COMPACT: one entry per model, in (iteration, slot) order (a dense
//                 [B * kMaxModels] response - 60 slots per iteration of the shared-focal solver, 0.2 of them used - was 3.8 MB of
//                 zero-filled vector per batch and problem)
//   kScore        score `seeds` -> counts, sums
//   kRefineScore  refine_model() of every seed and the scores of the results -> refined, rcounts, rsums
// counts / sums: inliers and the sum of their squared residuals in correspondence order, per model slot.  All decisions are
// taken here, in the reference's order: which hypotheses improve best_minimal_*, which of them seed a local optimisation
// (the last improving one of an iteration), the incumbent, the dynamic iteration bound and the stop rule.
// Traits: kSample (sample size), kMaxModels (model slots per iteration), finish(sum, count, N, options, focal) -> score_model().
template <class Traits> struct FocalLoop {
    static constexpr int kSample = Traits::kSample, kMaxModels = Traits::kMaxModels;
    enum Request { kDone = 0, kMinimal, kScore, kRefineScore };

    // ---- the request (valid after advance() returned it) ----
    uint64_t pos = 0;  // sampler draws consumed before the batch
    uint32_t B = 0;    // iterations of the batch
    std::vector<uint32_t> positions, prosac_samples;
    bool explicit_samples = false;
    std::vector<FocalModel> seeds;
    // ---- the response (filled by the caller before the next advance()) ----
    std::vector<FocalModel> models, refined;
    std::vector<uint32_t> num_models, counts, rcounts;
    std::vector<double> sums, rsums;

    FocalLoop(uint64_t N_, const FocalLoopOptions &o_, FocalModel *best_, FocalLoopStats *stats_)
        : N(N_), o(o_), best(best_), st(*stats_), dyn_max(o_.max_iterations), log_fail(std::log(1.0 - o_.success_prob)) {
        st = FocalLoopStats();
        if (o.progressive_sampling && N >= (uint64_t)kSample)
            prosac.init(o.seed, N, kSample, o.max_prosac_iterations);
    }

    Request advance() {
        for (;;) {
            switch (phase) {
            case Phase::kStart:
                if (N < (uint64_t)kSample) {
                    phase = Phase::kFinished;
                    return kDone;
                }
                if (o.score_initial_model) {
                    seeds.assign(1, *best);
                    phase = Phase::kInitialScored;
                    return kScore;
                }
                phase = Phase::kLoopHead;
                break;
            case Phase::kInitialScored: {
                const double sc = Traits::finish(sums[0], counts[0], N, o, best->f);
                const bool more = counts[0] > best_min_inl, better = sc < best_min_score;
                phase = Phase::kLoopHead;
                if (more || better) {
                    if (more)
                        best_min_inl = counts[0];
                    if (better)
                        best_min_score = sc;
                    if (sc < st.model_score) {
                        st.model_score = sc;
                        st.num_inliers = counts[0];
                    }
                    seeds.assign(1, *best);
                    phase = Phase::kInitialRefined;
                    return kRefineScore;
                }
                break;
            }
            case Phase::kInitialRefined:
                after_lo(refined[0], rcounts[0], rsums[0]);
                phase = Phase::kLoopHead;
                break;
            case Phase::kLoopHead: {
                // ransac_impl.h:109-111 at the head of the next iteration: the run may be over exactly at a batch boundary - do not
                // evaluate another batch to find that out
                if (stopped || st.iterations >= o.max_iterations || (st.iterations > o.min_iterations && st.iterations > dyn_max)) {
                    phase = Phase::kFinal;
                    break;
                }
                // the loop cannot stop before iteration max(min_iterations, dyn_max) + 1
                const uint64_t it0 = st.iterations;
                const uint64_t horizon = std::max(o.min_iterations, dyn_max) + 1;
                uint64_t want = horizon > it0 ? horizon - it0 : 1;
                // no model has bounded the run yet (dyn_max is still the iteration limit): evaluate up to the earliest possible stop,
                // then double - a typical run ends at min_iterations + 1, and a batch that is four times that only occupies the device
                // (the decisions do not depend on how the iterations are cut into batches)
                if (dyn_max >= o.max_iterations)
                    want = std::min<uint64_t>(want, std::max<uint64_t>(o.min_iterations + 1 > it0 ? o.min_iterations + 1 - it0 : 0, it0));
                want = std::min<uint64_t>(std::max<uint64_t>(want, 256), 4096);
                B = (uint32_t)std::min<uint64_t>(want, o.max_iterations - it0);
                positions.resize(B);
                pos_after = pos;
                explicit_samples = false;
                if (o.progressive_sampling) { // the subset-size recurrence is serial: B samples from the host
                    prosac_samples.resize((size_t)B * kSample);
                    for (uint32_t b = 0; b < B; ++b) {
                        positions[b] = 0;
                        prosac.generate(&prosac_samples[(size_t)b * kSample]);
                    }
                    explicit_samples = true;
                } else {
                    pos_after = focal_sample_positions<kSample>(o.seed, pos, N, B, positions.data());
                }
                phase = Phase::kBatchScored;
                return kMinimal;
            }
            case Phase::kBatchScored: {
                st.iterations_evaluated += B;
                // pass 1: the hypotheses that improve best_minimal_* (independent of the local optimisations)
                imps.clear();
                seeds.clear();
                size_t h = 0; // (compact index: the models of the batch in (iteration, slot) order)
                for (uint32_t b = 0; b < B; ++b) {
                    int last = -1;
                    for (uint32_t m = 0; m < num_models[b]; ++m, ++h) {
                        const double sc = Traits::finish(sums[h], counts[h], N, o, models[h].f);
                        const bool more = counts[h] > best_min_inl, better = sc < best_min_score;
                        if (!(more || better))
                            continue;
                        if (more)
                            best_min_inl = counts[h];
                        if (better)
                            best_min_score = sc;
                        imps.push_back(Improving{b, (uint32_t)h, counts[h], sc, -1});
                        last = (int)imps.size() - 1;
                    }
                    if (last >= 0) {
                        imps[last].job = (int)seeds.size();
                        seeds.push_back(models[imps[last].slot]);
                    }
                }
                phase = Phase::kBatchRefined;
                if (!seeds.empty()) // the local optimisations of the batch and the scores of their results: one device round trip
                    return kRefineScore;
                break;
            }
            case Phase::kBatchRefined: {
                // pass 2: the loop itself
                size_t a = 0;
                for (uint32_t b = 0; b < B; ++b) {
                    if (st.iterations > o.min_iterations && st.iterations > dyn_max) {
                        stopped = true;
                        break;
                    }
                    st.hypotheses += num_models[b];
                    for (; a < imps.size() && imps[a].iter == b; ++a) {
                        const Improving &im = imps[a];
                        if (im.score < st.model_score) {
                            st.model_score = im.score;
                            *best = models[im.slot];
                            st.num_inliers = im.count;
                        }
                        if (im.job >= 0)
                            after_lo(refined[im.job], rcounts[im.job], rsums[im.job]);
                    }
                    st.iterations++;
                }
                pos = pos_after;
                phase = Phase::kLoopHead;
                break;
            }
            case Phase::kFinal: // final polish (ransac_impl.h:190-198): model_score is not updated
                seeds.assign(1, *best);
                phase = Phase::kFinalRefined;
                return kRefineScore;
            case Phase::kFinalRefined: {
                st.refinements++;
                const double rsc = Traits::finish(rsums[0], rcounts[0], N, o, refined[0].f);
                if (rsc < st.model_score) {
                    *best = refined[0];
                    st.num_inliers = rcounts[0];
                }
                phase = Phase::kFinished;
                return kDone;
            }
            case Phase::kFinished:
                return kDone;
            }
        }
    }

  private:
    enum class Phase { kStart, kInitialScored, kInitialRefined, kLoopHead, kBatchScored, kBatchRefined, kFinal, kFinalRefined, kFinished };
    struct Improving {
        uint32_t iter, slot; // iteration of the batch, compact index of the model
        uint64_t count;
        double score;
        int job; // index into seeds / refined, -1: not the iteration's last improving hypothesis
    };
    const uint64_t N;
    const FocalLoopOptions o;
    FocalModel *const best;
    FocalLoopStats &st;
    Phase phase = Phase::kStart;
    uint64_t best_min_inl = 0;
    double best_min_score = std::numeric_limits<double>::max();
    uint64_t dyn_max;
    const double log_fail;
    uint64_t pos_after = 0;
    bool stopped = false;
    ProsacSampler prosac;
    std::vector<Improving> imps;

    // a local optimisation has returned: ransac_impl.h:124-154
    void after_lo(const FocalModel &ref, uint64_t rcnt, double rsum) {
        st.refinements++;
        const double rsc = Traits::finish(rsum, rcnt, N, o, ref.f);
        if (rsc < st.model_score) {
            st.model_score = rsc;
            st.num_inliers = rcnt;
            *best = ref;
        }
        st.inlier_ratio = static_cast<double>(st.num_inliers) / static_cast<double>(N);
        dyn_max = focal_dynamic_max_iter(st.num_inliers, N, kSample, log_fail, o.dyn_num_trials_mult, o.min_iterations, o.max_iterations);
    }
};

// One problem, one back end that evaluates every request synchronously:
//   int minimal(pos_base, positions, B, models, num_models [B], counts, sums, samples)   (models, counts, sums: compact)
//   int score(models, counts, sums)
//   int refine_score(seeds, refined, counts, sums)
template <class Traits, class Backend>
int focal_lo_ransac_t(Backend &be, uint64_t N, const FocalLoopOptions &o, FocalModel *best, FocalLoopStats *stats) {
    FocalLoop<Traits> loop(N, o, best, stats);
    for (;;) {
        int rc = 0;
        switch (loop.advance()) {
        case FocalLoop<Traits>::kDone:
            return 0;
        case FocalLoop<Traits>::kMinimal:
            rc = be.minimal(loop.pos, loop.positions.data(), loop.B, loop.models, loop.num_models, loop.counts, loop.sums,
                            loop.explicit_samples ? loop.prosac_samples.data() : nullptr);
            break;
        case FocalLoop<Traits>::kScore:
            rc = be.score(loop.seeds, loop.counts, loop.sums);
            break;
        case FocalLoop<Traits>::kRefineScore:
            rc = be.refine_score(loop.seeds, loop.refined, loop.rcounts, loop.rsums);
            break;
        }
        if (rc)
            return rc;
    }
}

// ransac_pnpf: FocalAbsolutePoseEstimator
struct PnpfTraits {
    static constexpr int kSample = kFocalSample, kMaxModels = kFocalMaxModels;
    static double finish(double inlier_sum, uint64_t count, uint64_t n, const FocalLoopOptions &o, double focal) {
        return focal_finish_score(inlier_sum, count, n, o.max_error, focal, o.max_focal);
    }
};
template <class Backend>
int focal_lo_ransac(Backend &be, uint64_t N, const FocalLoopOptions &o, FocalModel *best, FocalLoopStats *stats) {
    return focal_lo_ransac_t<PnpfTraits>(be, N, o, best, stats);
}

// ---- kernels (focal.hip) ----
struct FocalGenArgs {
    const double *a[5]; // x, y, X, Y, Z
    uint32_t n;
    uint64_t seed, pos_base;
    const uint32_t *positions;
    const uint32_t *samples; // optional: num_iters x kFocalSample explicit indices (PROSAC) instead of the counter-based draws
    uint32_t num_iters;
    double max_focal;     // < 0: no bound
    FocalModel *models;   // [num_iters * kFocalMaxModels]
    uint32_t *num_models; // [num_iters]
    double *stage;              // focal_stage_bytes(num_iters): workspace of the three generator kernels (focal.hip)
    const double *explicit_in;  // optional: num_iters x 20 minimal problems [x 4 x 2 | X 4 x 3] instead of samples of the points
    uint32_t keep_all;          // 1: every solution (the solver's interface), 0: the estimator's focal-length filter
    FocalModel *host_models;    // optional (pinned, mapped): the models and counts once more, for the host loop - no copy dispatches
    uint32_t *host_num_models;
};
struct FocalScoreArgs {
    const double *a[5];
    uint32_t n;
    const FocalModel *models;
    const uint32_t *num_models; // per group of kFocalMaxModels slots; nullptr: every slot holds a model
    const struct LMTask *lm_tasks; // optional: slot s = the pose and focal length k_lm_cam left in task s (instead of models)
    uint32_t num_slots;
    double thr2;
    uint32_t *counts; // [num_slots]  (slots without a model: 0)
    double *sums;     // [num_slots]
};

struct FocalMaskArgs { // get_inliers of one member of a group (k_focal_mask_g)
    const double *a[5];
    uint32_t n, pad;
    FocalModel model;
    double thr2;
    uint8_t *mask, *host_mask; // device mask (the final bundle reads it) and its pinned mirror (optional)
};

#if defined(__HIPCC__)
hipError_t launch_focal_generate_g(const FocalGenArgs *args, uint32_t G, uint32_t max_iters, hipStream_t stream);
hipError_t launch_focal_score_g(const FocalScoreArgs *args, uint32_t G, uint32_t max_slots, bool workgroup_per_model, hipStream_t stream);
hipError_t launch_focal_mask_g(const FocalMaskArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream);
hipError_t launch_focal_generate(const FocalGenArgs &g, hipStream_t stream);
hipError_t launch_focal_score(const FocalScoreArgs &a, hipStream_t stream);
size_t focal_stage_bytes(uint32_t num_iters);
hipError_t launch_focal_solve(const double *in, uint32_t count, FocalModel *models, uint32_t *num_models, double *stage,
                              uint32_t stage_samples, hipStream_t stream);
hipError_t launch_focal_mask(const double *const *a, uint32_t n, const FocalModel &m, double thr2, uint8_t *mask, uint8_t *host_mask,
                             hipStream_t stream);
#endif

} // namespace pl
