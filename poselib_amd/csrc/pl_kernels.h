// poselib_amd — launch interface between the host driver (driver.cc) and the HIP kernels
// (kernels.hip).  Plain structs and pointers only.
#pragma once
#include "pl_prefilter.h"
#include "pl_refine.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pl {

// Correspondences of one problem, structure-of-arrays in HBM, fp64.
//   absolute pose : a[0..4] = x, y (normalised image plane), X, Y, Z
//   two-view      : a[0..3] = x1, y1, x2, y2
struct PointSet {
    const double *a[5];
    uint32_t n;
    float xy_absmax; // absolute pose: max(|x|, |y|) over the 2-D points; two-view: max over all four coordinates, +inf
                     // when unknown (bounds used by the scoring pre-filters)
};

constexpr int kScoreThreads = 256; // 4 wavefronts per workgroup
constexpr int kLMSeqPoints = 256; // up to this many correspondences k_lm sums in the reference's order (one thread per correspondence)
#ifndef PL_LM_THREADS
#define PL_LM_THREADS 512
#endif
constexpr int kLMThreads = PL_LM_THREADS; // k_lm: wavefront 0 adds the term rows in order, the others produce them (kernels.hip)

PL_HD constexpr int sample_size(int est) { return est == EST_ABS ? 3 : est == EST_REL ? 5 : est == EST_FUND ? 7 : 4; }
PL_HD constexpr int max_models(int est) { return est == EST_ABS ? 4 : est == EST_REL ? 40 : est == EST_FUND ? 3 : 1; }
PL_HD constexpr int point_doubles(int est) { return est == EST_ABS ? 5 : 4; }

struct BatchCtl;
struct GenerateArgs {
    PointSet pts;
    uint64_t seed;
    uint64_t pos_base;         // draws consumed before the batch
    const uint32_t *positions; // draws consumed before each iteration, relative to pos_base
    const uint32_t *samples;   // optional explicit minimal samples [num_iters][K] (PROSAC: drawn on the host,
                               // the subset schedule is a serial recurrence); nullptr => draw from `positions`
    uint32_t num_iters;
    uint32_t slots_per_iter;   // record slots reserved per iteration (<= max_models(est)); more solutions than
                               // slots => ctl->gen_overflow is set and the host repeats the batch with more room
    struct BatchCtl *ctl;
    double *models;            // [num_iters * slots_per_iter] records of kModelStride doubles
    uint32_t *num_models;      // [num_iters]
    int32_t real_focal_check;  // fundamental only
    uint32_t *blk_tot = nullptr; // optional, zeroed: models per block of 1024 iterations, accumulated by the generator
    uint32_t *blk_nan = nullptr; // optional, zeroed: NaN models per block of 1024 iterations (statistics)
    void *stage = nullptr;     // relative pose: workspace of generate_stage_bytes(); nullptr = single-kernel generator
};

struct ScoreArgs {
    PointSet pts;
    const double *models;
    const uint32_t *slots;     // hypothesis k -> model record index (nullptr: identity)
    const float *shadow;       // optional compact [num_hyp][16] fp32 model shadows in hypothesis order (pre-filter)
    const double *compact64;   // optional compact [num_hyp][16] fp64 model fields in hypothesis order (pre-filter)
    const void *shadow16;      // optional fp16 MFMA operand blocks of the hypotheses (absolute pose, k_score_mfma):
                               // 96 B per hypothesis in groups of 16, see k_shadow16 in pipeline.hip
    const uint32_t *num_hyp;   // device scalar
    uint32_t hyp_capacity;     // row pitch of the partial arrays
    double thr2;
    PrefilterArgs pf;          // conservative fp32 pre-filter (pl_prefilter.h); pf.enabled == 0: exact evaluation
    uint32_t *part_count;      // [chunks][hyp_capacity]
    double *part_score;        // [chunks][hyp_capacity]
    uint32_t *tickets;         // [chunks], zeroed before the launch: work counters of the waves that share a chunk
};
constexpr uint32_t kMaxScoreChunks = 4096; // (2^31 correspondences / 64 P would be more; the driver rejects such sets)

struct FinalizeArgs {
    const uint32_t *num_hyp;
    uint32_t hyp_capacity, chunks, n_points;
    double thr2;
    const uint32_t *part_count;
    const double *part_score;
    uint32_t *count; // [hyp_capacity]
    double *score;   // [hyp_capacity]
};

struct LMTask {
    PointSet pts;                 // the correspondences the task refines on
    double params[kParamDoubles]; // in/out (see pl_refine.h for the per-problem layout)
    LMOptions opt;
    CameraParams cam;      // absolute pose only
    double point_scale;    // absolute pose: 2-D points are multiplied by this on load (final bundle)
    double prefilter_thr2; // relative pose LO: > 0 => keep only points passing Sampson+cheirality at this
                           // threshold; skip the refinement when <= 5 survive (relative_pose.cc:62-86)
    const uint8_t *mask;   // optional inlier mask (final polish on inliers); nullptr = all points
    uint8_t *scratch;      // n bytes, used by the prefilter
    const double *record_in; // k_lm: the seed's model record (kept when the refinement is skipped) ...
    double *record_out;      // ... and where the refined model's record goes (device memory); nullptr: no record
    int32_t cam_flags;       // k_lm_cam: CamRefineFlags (pl_refine_cam.h) - the intrinsics refined with the pose; `cam` is in/out
    int32_t pad0;
    // A task that is enqueued BEHIND the kernels that decide whether and from where it runs (pl_estimate_batch: the front-end's final
    // bundle behind the final refinement, its choice and the inlier mask - one synchronisation instead of two):
    const double *start_record;  // optional: the model record (device memory) whose parameters replace `params` as the starting point
    const uint32_t *gate_count;  // optional: the task runs only if *gate_count > gate_min (robust.cc:103 "num_inliers > 3" etc.); otherwise
    uint32_t gate_min;           // skipped = 2 and nothing else is written
    uint32_t pad1;
    // outputs
    uint32_t iterations, skipped;
    double cost, initial_cost;
};

// Device-resident control block of one batch (zeroed before the batch, read back after it).
struct BatchCtl {
    uint32_t num_hyp;     // hypotheses of the batch (k_compact2)
    uint32_t num_records; // improving hypotheses found by k_records (may exceed the list capacity)
    uint32_t orbit_error; // sampler position window too small / too many redraw segments
    uint32_t gen_overflow; // an iteration produced more models than slots_per_iter
    uint64_t pos_after;   // draws consumed after the batch's last iteration
    uint32_t nan_hyp;     // hypotheses of the batch with a NaN entry (counted by the generators, summed by k_compact2;
                          // the scorers skip them: no inliers)
    uint32_t pad;
};
struct RecordMeta {
    uint32_t k, slot, count, pad;
    double score;
};

// k_score_seq: scores in the reference's (sequential) summation order for the models decisions are taken on.
struct SeqScoreArgs {
    PointSet pts;
    const double *models;   // records of kModelStride doubles
    RecordMeta *cand;       // candidate list: model of candidate r = models + cand[r].slot; count / score are rewritten.
                            // nullptr: model r = models + r, results go to count[] / score[]
    const uint32_t *num;    // number of candidates / models (device memory)
    uint32_t cap;           // upper bound of *num
    double thr2;
    uint32_t *count;        // plain mode: [cap]
    double *score;          // plain mode: [cap]
    uint32_t *host_count;   // plain mode: optional pinned mirrors
    double *host_score;
    RecordMeta *host_cand;  // candidate mode: pinned mirror of the first host_cap candidates
    uint32_t host_cap;
    const BatchCtl *ctl_src = nullptr; // optional: the batch's control block, final when this kernel starts, is copied
    BatchCtl *ctl_host = nullptr;      // to pinned host memory by the kernel itself (no copy dispatch)
};
hipError_t launch_score_seq(int est, const SeqScoreArgs &a, hipStream_t stream);

// All launchers enqueue on `stream` and return the HIP status of the launch.
hipError_t launch_generate(int est, const GenerateArgs &a, hipStream_t stream);
size_t generate_stage_bytes(int est, uint32_t num_iters); // workspace of the staged 5-point generator (0: none)
// the staged 5-point generator (gen_rel.hip): a.stage = workspace of rel_stage_bytes(a.num_iters)
size_t rel_stage_bytes(uint32_t num_iters);
hipError_t launch_generate_rel(const GenerateArgs &a, hipStream_t stream);
struct GroupArgs;
hipError_t launch_group_generate_rel(const GroupArgs *args, uint32_t max_B, uint32_t G, hipStream_t stream);
// Front-end pre-processing on the device (robust.cc:40-46, 286-292; utils.cc:584-644 per-point part): AoS user
// buffers -> the problem's SoA block, with per-point un-projection (modes 0, 1) or the affine normalisation whose
// centroid / scale the host has summed sequentially (mode 2; the two reductions of normalize_points are order
// dependent and stay on the host to remain bit-identical).  absmax_bits: max(|x|, |y|) of the first point set as the
// bit pattern of a non-negative double (NaN propagates), zeroed by the caller.
struct PrepareArgs {
    int32_t mode;          // 0: a -> unproject(cam1), b = N x 3 copied; 1: a -> unproject(cam1), b -> unproject(cam2);
                           // 2: a -> (a - c1) / scale, b -> (b - c2) / scale   (subtract only if `centred`)
    int32_t centred;
    CameraParams cam1, cam2;
    double c1x, c1y, c2x, c2y, scale;
};
hipError_t launch_prepare(const double *a_raw, const double *b_raw, uint32_t n, const PrepareArgs &args, double *soa,
                          unsigned long long *absmax_bits, hipStream_t stream);
// pixels of a distorting camera -> pixels of the distortion-free camera with the same focal lengths / principal point
// diagnostic: fn 0 cube of the LM's Nielsen update, 1 sqrt, 2 reciprocal, 3 cbrt, 4 cos, 5 sin, 6 acos - as the kernels evaluate them
hipError_t launch_device_math(int fn, const double *x, uint32_t n, double *out, hipStream_t stream);
hipError_t launch_undistort(const double *in, uint32_t n, const CameraParams &cam, double fx, double fy, double cx,
                            double cy, double *out, hipStream_t stream);
// After the LM kernels: records of the refined models on the device (skipped tasks keep their input record), and the
// choice "refined model if its score beats `incumbent_score`, else the incumbent" of the final refinement
// (ransac_impl.h:190-198) so that the inlier mask can follow without a host round trip.
// host_tasks (pinned host memory, may be null): receives params / skipped / iterations / costs of every task
hipError_t launch_task_records(int est, const LMTask *tasks, const double *records_in, double *records_out,
                               uint32_t num_tasks, LMTask *host_tasks, hipStream_t stream);
hipError_t launch_select_record(const double *score_refined, double incumbent_score, const double *rec_refined,
                                const double *rec_incumbent, double *out, hipStream_t stream);
// fp16 A-operand blocks for k_score_mfma from the compact fp32 shadows (absolute pose); capacity = hypotheses rounded
// up to a multiple of 8, 64 B each.
hipError_t launch_shadow16(const uint32_t *num_hyp, const float *shadow_compact, uint32_t hyp_capacity, float g16,
                           float c16, float thr, void *shadow16, hipStream_t stream);
bool score_uses_mfma(int est, uint32_t n_points, const PrefilterArgs &pf);
// operands of k_score_mfma2 for `capacity` hypotheses listed in `slots` (+ the pad rows behind them)
hipError_t launch_sampson16(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t capacity, void *out,
                            hipStream_t stream);
// operands of k_score_mfmah for `capacity` hypotheses listed in `slots` (the last group of 8 is filled up)
hipError_t launch_hom16(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t capacity, float thr, void *out,
                        hipStream_t stream);
size_t lm2_state_bytes(uint32_t num_tasks);
size_t lm2_partial_bytes(uint32_t num_tasks, uint32_t slices);
hipError_t launch_lm2(int est, const PointSet &pts, LMTask *tasks, uint32_t num_tasks, uint32_t slices,
                      uint32_t max_iterations, void *states, double *partials, hipStream_t stream);
// chunks = ceil(n / (kScoreThreads * P)); P is chosen inside from n (returned through *chunks_out)
uint32_t score_chunks(int est, uint32_t n_points, bool prefilter, bool mfma = false);
hipError_t launch_score(int est, const ScoreArgs &a, uint32_t slices, hipStream_t stream);
// num_models[iters] -> slots (compact list of record indices in (iteration, model) order) + count
hipError_t launch_lm(int est, const PointSet &pts, LMTask *tasks, uint32_t num_tasks, hipStream_t stream);
// host_mask (pinned, device-mapped; may be null): the kernel writes a second copy there itself - no copy dispatch
hipError_t launch_mask(int est, const PointSet &pts, const double *model, double thr2, uint8_t *mask,
                       uint8_t *host_mask, hipStream_t stream);

// ---- device-side bookkeeping (pipeline.hip) ----
// zero_words: 32-bit words starting at `ctl` (control block + the generators' per-block model counts) that the first
// kernel zeroes itself (0: the caller has done it)
hipError_t launch_sample_positions(int K, uint64_t seed, uint64_t pos_base, uint64_t N, uint32_t B, uint32_t M,
                                   uint8_t *delta, uint64_t *flagbits /* >= ceil(M / 64) words */, uint32_t *positions,
                                   BatchCtl *ctl, uint32_t zero_words, hipStream_t stream);
// counted: blk_tot already holds the per-block model counts (GenerateArgs.blk_tot); otherwise they are counted first
// shadow16 != nullptr: the fp16 operand blocks of k_score_mfma are built in the same launch as the hypothesis-ordered
// copies (Shadow16Params; see k_shadow16)
struct Shadow16Params {
    void *out = nullptr;
    float g16 = 0.f, c16 = 0.f, thr = 0.f;
    int sampson = 0; // 1: two-view, operands of k_score_mfma2 (96 B per hypothesis, pl_prefilter.h Sampson16Operand);
                     // 2: homography, operands of k_score_mfmah (256 B per hypothesis, Hom16Model; thr = PrefilterArgs.h16)
};
constexpr size_t kSampson16Bytes = 96;
constexpr size_t kAbs16Bytes = 64;  // absolute pose (k_score_mfma): 3 directions + the shared second k block, 16 B each, per hypothesis
constexpr size_t kAbs16Pad = 32;    // the last group of 32 is filled up
constexpr size_t kSampson16Pad = 64; // operand rows a partial group of 32 may read past the last hypothesis
constexpr size_t kHom16Bytes = 256; // homography (k_score_mfmah): 4 rows x 4 k blocks x 16 B per hypothesis, groups of 8
constexpr size_t kHom16Pad = 8;     // the last group of 8 is filled up
// blk_tot is followed by the generators' NaN-model table of the same length (nb = ceil(B / 1024) entries each)
hipError_t launch_compact2(const uint32_t *num_models, uint32_t B, int maxm, uint32_t *blk_tot, bool counted, uint32_t *slots,
                           uint32_t *offsets, const double *models, float *shadow_compact, double *compact64,
                           BatchCtl *ctl, const Shadow16Params &s16, hipStream_t stream);
// the two halves on their own (diagnostic entry pl_debug_score_stream): hypothesis-ordered copies of records that already
// sit in `models` in list order `slots`, and the chunk partials -> (count, score) without the record scan
hipError_t launch_gather_models(BatchCtl *ctl, const uint32_t *slots, const double *models, uint32_t capacity,
                                float *shadow_compact, double *compact64, hipStream_t stream);
hipError_t launch_finalize(const FinalizeArgs &f, hipStream_t stream);
hipError_t launch_finalize_records(const FinalizeArgs &f, const uint32_t *slots, const double *models,
                                   uint32_t *blk_max, double *blk_min, uint32_t init_max, double init_min,
                                   RecordMeta *rec_meta, double *rec_models, uint32_t rec_cap, BatchCtl *ctl,
                                   RecordMeta *host_meta, double *host_models, uint32_t host_cap, hipStream_t stream);

// ---- groups of problems: one launch sequence for MANY independent problems (BASELINE config 4) ----------------------
// The problem index is a grid dimension (blockIdx.z) and every kernel fetches its arguments from a device-resident
// table with one entry per problem of the group - the same kernel bodies as the single-problem launches above, so a
// batch of 64 default-option problems costs the ~10 launches of one problem instead of 64 x 10.
struct SampleArgs {
    uint64_t seed, pos_base, N;
    uint32_t B, M;
    uint8_t *delta;
    uint64_t *flagbits;
    uint32_t *positions;
    BatchCtl *ctl;
    uint32_t zero_words, pad;
};
struct CompactArgs {
    const uint32_t *num_models;
    uint32_t B;
    int32_t maxm;
    uint32_t *blk_tot, *slots, *offsets;
    const double *models;
    float *shadow;
    double *compact64;
    BatchCtl *ctl;
    Shadow16Params s16;
    uint32_t *host_offsets; // optional pinned mirror of `offsets` (the host needs offsets[stop] after its replay)
};
struct RecordsArgs {
    FinalizeArgs f;
    const uint32_t *slots;
    const double *models;
    uint32_t *blk_max;
    double *blk_min;
    uint32_t init_max, rec_cap;
    double init_min;
    RecordMeta *rec_meta;
    double *rec_models;
    BatchCtl *ctl;
    RecordMeta *host_meta;
    double *host_models;
    uint32_t host_cap, pad;
};
struct MaskArgs {
    PointSet pts;
    const double *model;
    double thr2;
    uint8_t *mask, *host_mask;
};
struct SelectArgs {
    const double *score_refined;
    double incumbent_score;
    const double *rec_refined, *rec_incumbent;
    double *out;
    const uint32_t *count_refined; // optional: inlier count of the refined model ...
    uint32_t count_incumbent;      // ... and of the incumbent; the chosen one goes to
    uint32_t pad;
    uint32_t *count_out;           // ... this device word (gate of the tasks enqueued behind the choice)
    const uint32_t *fetch_src; // optional: one device word (a stopped problem's hypothesis offset) ...
    uint32_t *fetch_dst;       // ... copied to this (pinned host) address by the same launch
};
struct GroupArgs { // everything the kernels of one batch step need for ONE problem of the group
    uint32_t active;   // 0: the slot takes no part in this step
    uint32_t use_mfma; // absolute pose: scored by k_score_mfma (otherwise k_score_queue)
    uint32_t chunks, slices;
    SampleArgs samp;
    GenerateArgs gen;
    CompactArgs comp;
    ScoreArgs score;
    RecordsArgs rec;
    SeqScoreArgs seq;
};
// chunk = 64 * this many correspondences for two-view problems scored by k_score_mfma2 in a group (score_shape's bound)
constexpr int group_mfma2_points_per_lane(int est) { return est == EST_REL ? 5 : 6; }
struct GroupDims { // grid extents: maxima over the active problems of the group
    uint32_t G;          // problems (grid.z)
    uint32_t max_M;      // sampler window
    uint32_t min_n;      // fewest correspondences (most sample redraws: sizes the orbit kernel's tables)
    uint32_t max_B;      // iterations per batch
    uint32_t max_hcap;   // hypothesis slots
    uint32_t max_chunks, max_slices;
    uint32_t any_mfma, any_queue;
    int P;               // points per lane of the scorers (chunk = 64 P correspondences)
};
// positions -> generate -> compact/gather(/fp16 operands) -> score -> finalize/records -> candidates re-scored: the
// whole batch step of every active problem of the group, one launch per kernel
// ev0 / ev1 (optional): recorded around the scoring launch(es)
hipError_t launch_group_batch(int est, const GroupArgs *args, const GroupDims &dims, hipStream_t stream,
                              hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t launch_group_score_seq(int est, const SeqScoreArgs *args, uint32_t G, uint32_t max_cap, hipStream_t stream);
hipError_t launch_group_select(const SelectArgs *args, uint32_t G, hipStream_t stream);
hipError_t launch_group_mask(int est, const MaskArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream);
struct PrepareGroupArgs {
    const double *a_raw, *b_raw;
    uint32_t n, pad;
    PrepareArgs args;
    double *soa;
    unsigned long long *absmax_bits;
};
hipError_t launch_group_prepare(const PrepareGroupArgs *args, uint32_t G, uint32_t max_n, hipStream_t stream);
// (pieces of launch_group_batch that live in pipeline.hip)
hipError_t launch_group_positions(int K, const GroupArgs *args, const GroupDims &d, hipStream_t stream);
hipError_t launch_group_compact(const GroupArgs *args, const GroupDims &d, hipStream_t stream);
hipError_t launch_group_finalize_records(const GroupArgs *args, const GroupDims &d, hipStream_t stream);
// LM over the tasks of many problems: every task carries its own correspondences (LMTask.pts)
hipError_t launch_lm_tasks(int est, LMTask *tasks, uint32_t num_tasks, uint32_t max_points, hipStream_t stream);
// Summation order of the refinements.  0 (default): k_lm - reference order up to 256 correspondences, tree beyond - for poses and
// homographies, k_lm_ordered for fundamental matrices (the sign of a refined F hangs on the bits of its input, pl_svd3.h);
// 1: k_lm_ordered for every estimator; 2: k_lm for every estimator
void set_lm_mode(int mode);
int get_lm_mode();
bool lm_sums_ordered(int est); // what launch_lm_tasks will pick for this estimator
void set_lm_force_ordered(int on); // this host thread's LM launches in the reference's order whatever the mode (0: back to the mode)
// absolute pose + camera intrinsics (lm_cam.hip): tasks with cam_flags != 0; refined pose -> params / record_out, camera -> cam
hipError_t launch_lm_cam(LMTask *tasks, uint32_t num_tasks, hipStream_t stream);
int group_points_per_lane(int est); // P of the group launches (fixed per estimator)

// Bare solver entry points (one problem per lane); inputs/outputs in HBM.
//   abs : in = [x0 x1 x2 X0 X1 X2] (18 doubles / problem) -> out records (4 / problem)
hipError_t launch_solve_batch(int est, const double *in, uint32_t num_problems, double *models, uint32_t *num_models,
                              hipStream_t stream);

} // namespace pl
