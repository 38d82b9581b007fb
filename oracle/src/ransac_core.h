// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the LO-RANSAC control loop, its option/stat records and the sampler.
//   sampler      : PoseLib/robust/sampling.cc:37-61 (splitmix64, rejection sampling),
//                  :85-136 (PROSAC schedule)
//   loop         : PoseLib/robust/ransac_impl.h:43-73 (iteration bound), :99-154 (per-iteration
//                  bookkeeping + LO trigger), :157-201 (outer loop, stop rule, final refine)
//   option/stats : PoseLib/types.h:39-58
// Loop control is pinned by the reference's own tests/ransac_test.cc:38-122 (see
// tests/test_oracle_ransac_control.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace orc {

struct RansacOptions { // types.h:39-50
    uint64_t max_iterations = 100000;
    uint64_t min_iterations = 1000;
    double dyn_num_trials_mult = 3.0;
    double success_prob = 0.9999;
    uint64_t seed = 0;
    bool progressive_sampling = false;
    uint64_t max_prosac_iterations = 100000;
    bool score_initial_model = false;
};

struct RansacStats { // types.h:52-58
    uint64_t refinements = 0;
    uint64_t iterations = 0;
    uint64_t num_inliers = 0;
    double inlier_ratio = 0;
    double model_score = std::numeric_limits<double>::max();
};

// ------------------------------------------------------------------------- sampler
// One splitmix64 step.  The reference returns `int`: the 64-bit hash is truncated to its
// low 32 bits and reinterpreted as signed (sampling.cc:37-43); the caller then converts
// to size_t (sign extension) before `% N` (sampling.cc:50).
inline int32_t splitmix_next_int(uint64_t &state) {
    state += 0x9e3779b97f4a7c15ULL;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z = z ^ (z >> 31);
    return static_cast<int32_t>(static_cast<uint32_t>(z));
}
inline uint64_t draw_index(uint64_t &state, uint64_t N) {
    const int32_t r = splitmix_next_int(state);
    return static_cast<uint64_t>(static_cast<int64_t>(r)) % N;
}
// K distinct indices in [0,N); duplicates are redrawn (each redraw consumes RNG state).
inline void draw_distinct(uint64_t &state, uint64_t N, size_t K, uint64_t *out) {
    for (size_t i = 0; i < K; ++i) {
        for (;;) {
            out[i] = draw_index(state, N);
            bool fresh = true;
            for (size_t j = 0; j < i; ++j)
                if (out[j] == out[i]) {
                    fresh = false;
                    break;
                }
            if (fresh)
                break;
        }
    }
}

class Sampler { // sampling.h:49-82
  public:
    Sampler(size_t N, size_t K, const RansacOptions &opt)
        : num_data(N), sample_sz(K), state(opt.seed), prosac(opt.progressive_sampling),
          prosac_limit(opt.max_prosac_iterations) {
        if (prosac)
            init_prosac();
    }
    void next(uint64_t *sample) { // sampling.cc:85-103
        if (prosac && sample_k < prosac_limit) {
            draw_distinct(state, subset_sz - 1, sample_sz - 1, sample);
            sample[sample_sz - 1] = subset_sz - 1;
            ++sample_k;
            if (sample_k < prosac_limit && sample_k > growth[subset_sz - 1]) {
                if (++subset_sz > num_data)
                    subset_sz = num_data;
            }
        } else {
            draw_distinct(state, num_data, sample_sz, sample);
        }
    }
    size_t num_data, sample_sz;
    uint64_t state;

  private:
    void init_prosac() { // sampling.cc:105-136
        growth.assign(std::max(num_data, sample_sz), 0);
        double Tn = static_cast<double>(prosac_limit);
        for (size_t i = 0; i < sample_sz; ++i)
            Tn *= static_cast<double>(sample_sz - i) / (num_data - i);
        for (size_t n = 0; n < sample_sz; ++n)
            growth[n] = 1;
        uint64_t Tn_prime = 1;
        for (size_t n = sample_sz; n < num_data; ++n) {
            const double Tn_next = Tn * (n + 1.0) / (n + 1.0 - sample_sz);
            Tn_prime += std::ceil(Tn_next - Tn);
            growth[n] = Tn_prime;
            Tn = Tn_next;
        }
        sample_k = 1;
        subset_sz = sample_sz;
    }
    bool prosac;
    uint64_t prosac_limit;
    uint64_t sample_k = 0, subset_sz = 0;
    std::vector<uint64_t> growth;
};

// ------------------------------------------------------------------------- iteration bound
inline double prob_all_inlier_sample(uint64_t inl, uint64_t N, uint64_t K) { // ransac_impl.h:43-56
    if (K == 0)
        return 1.0;
    if (inl < K || N < K)
        return 0.0;
    double p = 1.0;
    for (uint64_t i = 0; i < K; ++i)
        p *= static_cast<double>(inl - i) / static_cast<double>(N - i);
    return p;
}
inline uint64_t dynamic_iteration_bound(uint64_t inl, uint64_t N, uint64_t K, double log_fail, double mult,
                                        uint64_t min_it, uint64_t max_it) { // ransac_impl.h:58-73
    const double p = prob_all_inlier_sample(inl, N, K);
    if (p >= 0.9999)
        return min_it;
    if (p <= 0.0001)
        return max_it;
    const uint64_t n = static_cast<uint64_t>(std::ceil(log_fail / std::log(1.0 - p) * mult));
    return std::max(min_it, std::min(max_it, n));
}

// ------------------------------------------------------------------------- loop
struct LoopState { // ransac_impl.h:99-104
    uint64_t best_min_inliers = 0;
    double best_min_score = std::numeric_limits<double>::max();
    uint64_t dyn_max_iter = 100000;
    double log_fail = 0;
};

// Optional trace of the loop (used by tests to compare the GPU replay against the oracle).
struct LoopTrace {
    uint64_t hypotheses = 0;               // minimal models passed to score() in the main loop
    std::vector<uint64_t> trigger_iters;   // iterations that ran LO
};

// Est must provide: sample_sz, num_data, generate(std::vector<Model>*), score(const Model&, uint64_t*),
// refine(Model*).
template <typename Est, typename Model>
void process_candidates(Est &est, const std::vector<Model> &cands, const RansacOptions &opt, LoopState &st,
                        RansacStats &stats, Model *best) { // ransac_impl.h:106-154
    int lo_seed = -1;
    uint64_t cnt = 0;
    for (size_t i = 0; i < cands.size(); ++i) {
        const double sc = est.score(cands[i], &cnt);
        const bool more = cnt > st.best_min_inliers;
        const bool better = sc < st.best_min_score;
        if (!(more || better))
            continue;
        if (more)
            st.best_min_inliers = cnt;
        if (better)
            st.best_min_score = sc;
        lo_seed = static_cast<int>(i);
        if (sc < stats.model_score) {
            stats.model_score = sc;
            *best = cands[i];
            stats.num_inliers = cnt;
        }
    }
    if (lo_seed < 0)
        return;

    Model refined = cands[lo_seed];
    est.refine(&refined);
    stats.refinements++;
    const double rsc = est.score(refined, &cnt);
    if (rsc < stats.model_score) {
        stats.model_score = rsc;
        stats.num_inliers = cnt;
        *best = refined;
    }
    stats.inlier_ratio = static_cast<double>(stats.num_inliers) / static_cast<double>(est.num_data);
    st.dyn_max_iter = dynamic_iteration_bound(stats.num_inliers, est.num_data, est.sample_sz, st.log_fail,
                                              opt.dyn_num_trials_mult, opt.min_iterations, opt.max_iterations);
}

template <typename Est, typename Model>
RansacStats lo_ransac(Est &est, const RansacOptions &opt, Model *best, LoopTrace *trace = nullptr) { // :157-201
    RansacStats stats;
    if (est.num_data < est.sample_sz)
        return stats;
    stats.num_inliers = 0;
    stats.model_score = std::numeric_limits<double>::max();
    LoopState st;
    st.dyn_max_iter = opt.max_iterations;
    st.log_fail = std::log(1.0 - opt.success_prob);

    std::vector<Model> cands;
    if (opt.score_initial_model) {
        cands.push_back(*best);
        process_candidates(est, cands, opt, st, stats, best);
    }
    for (stats.iterations = 0; stats.iterations < opt.max_iterations; stats.iterations++) {
        if (stats.iterations > opt.min_iterations && stats.iterations > st.dyn_max_iter)
            break;
        cands.clear();
        est.generate(&cands);
        const uint64_t before = stats.refinements;
        process_candidates(est, cands, opt, st, stats, best);
        if (trace) {
            trace->hypotheses += cands.size();
            if (stats.refinements != before)
                trace->trigger_iters.push_back(stats.iterations);
        }
    }
    // final polish: note that model_score is deliberately NOT updated here (:195-198)
    Model refined = *best;
    est.refine(&refined);
    stats.refinements++;
    uint64_t cnt = 0;
    const double rsc = est.score(refined, &cnt);
    if (rsc < stats.model_score) {
        *best = refined;
        stats.num_inliers = cnt;
    }
    return stats;
}

} // namespace orc
