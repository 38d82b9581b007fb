"""Oracle pins that come from the reference itself.

* loop control: the four cases of the reference's tests/ransac_test.cc:38-122, re-expressed against the
  oracle's loop (MockEstimator: one dummy model per iteration, fixed inlier count, score 0);
* sampler: known-answer vectors of sampling.cc:37-61.
"""
import math

import numpy as np

import oracle_lib as O


def expected_iterations(dyn, mn, mx):  # ransac_test.cc:30-36
    stop_after = max(mn, dyn)
    return mx if stop_after >= mx else stop_after + 1


def test_all_inlier_probability_is_hypergeometric():  # ransac_test.cc:38-51
    expected = (5 / 10) * (4 / 9) * (3 / 8) * (2 / 7) * (1 / 6)
    got = O.lib().orc_all_inlier_probability(5, 10, 5)
    assert abs(got - expected) < 1e-12
    assert abs(got - (5 / 10) ** 5) > 1e-3
    assert O.lib().orc_all_inlier_probability(4, 10, 5) == 0.0
    assert O.lib().orc_all_inlier_probability(8, 8, 5) == 1.0
    assert O.lib().orc_all_inlier_probability(3, 10, 0) == 1.0


def test_dynamic_iterations_use_exact_probability():  # ransac_test.cc:53-77
    opt = dict(min_iterations=0, max_iterations=1000, dyn_num_trials_mult=1.0, success_prob=0.5)
    p = (5 / 10) * (4 / 9) * (3 / 8) * (2 / 7) * (1 / 6)
    dyn = math.ceil(math.log(1 - 0.5) / math.log(1 - p) * 1.0)
    assert dyn == 175
    assert O.lib().orc_dynamic_max_iter(5, 10, 5, math.log(0.5), 1.0, 0, 1000) == dyn
    st = O.mock_ransac(10, 5, 5, opt)
    assert st["num_inliers"] == 5
    assert st["iterations"] == expected_iterations(dyn, 0, 1000) == 176


def test_dynamic_iterations_stay_at_max_when_not_enough_inliers():  # ransac_test.cc:79-99
    opt = dict(min_iterations=0, max_iterations=20, dyn_num_trials_mult=1.0, success_prob=0.5)
    assert O.lib().orc_dynamic_max_iter(4, 10, 5, math.log(0.5), 1.0, 0, 20) == 20
    st = O.mock_ransac(10, 5, 4, opt)
    assert st["num_inliers"] == 4 and st["iterations"] == 20


def test_dynamic_iterations_collapse_to_min_for_all_inliers():  # ransac_test.cc:101-122
    opt = dict(min_iterations=3, max_iterations=100, dyn_num_trials_mult=1.0, success_prob=0.5)
    assert O.lib().orc_dynamic_max_iter(8, 8, 5, math.log(0.5), 1.0, 3, 100) == 3
    st = O.mock_ransac(8, 5, 8, opt)
    assert st["num_inliers"] == 8 and st["iterations"] == expected_iterations(3, 3, 100) == 4


SAMPLER_KATS = {  # (N, K) -> first three samples, seed 0
    (5000, 3): [[767, 1356, 535], [1620, 4395, 1298], [3705, 364, 2299]],
    (5000, 5): [[767, 1356, 535, 1620, 4395], [1298, 3705, 364, 2299, 4926], [3457, 1934, 2123, 551, 4081]],
    (10000, 4): [[767, 6356, 5535, 6620], [4395, 6298, 8705, 364], [7299, 9926, 8457, 6934]],
    (10000, 7): [[767, 6356, 5535, 6620, 4395, 6298, 8705], [364, 7299, 9926, 8457, 6934, 2123, 5551],
                 [9081, 7419, 2917, 6006, 2660, 108, 7015]],
}


def test_sampler_known_answers():
    for (N, K), want in SAMPLER_KATS.items():
        idx, state = O.sampler_draw(0, N, K, 3)
        assert idx.tolist() == want
        if (N, K) == (5000, 3):
            assert state == 0x8FF34785799E5CBD


def test_sampler_draw_accounting():
    # 100 000 P3P iterations at N=5000 consume 300 063 draws (63 duplicate redraws)
    _, state = O.sampler_draw(0, 5000, 3, 100000)
    G = 0x9E3779B97F4A7C15
    draws = (state * pow(G, -1, 2**64)) % 2**64
    assert draws == 300063
    # half of the raw draws are negative ints (the int truncation + sign extension is part of the contract)
    st = O.C.c_uint64(0)
    neg = sum(1 for _ in range(100000) if O.lib().orc_random_int(O.C.byref(st)) < 0)
    assert 49000 < neg < 51000


def test_sampler_indices_distinct_and_in_range():
    for N, K in [(7, 7), (10, 5), (200, 3), (5000, 5)]:
        idx, _ = O.sampler_draw(12345, N, K, 2000)
        assert idx.max() < N
        assert all(len(set(r)) == K for r in idx.tolist())


def test_prosac_prefix_growth():
    # PROSAC draws the last index as subset_sz-1 and grows the subset (sampling.cc:85-100)
    idx, _ = O.sampler_draw(1, 1000, 4, 200, prosac=True, max_prosac=100000)
    assert idx[0, 3] == 3 and (idx[0, :3] < 3).all()
    assert (np.diff(idx[:, 3].astype(np.int64)) >= 0).all()
