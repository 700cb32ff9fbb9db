"""Reproducible cross-chain sums on the CPU: the oracle's restatement against an independent
big-integer model, order / partition independence, and the host-side record arithmetic of libamc.so (amc_xsum_merge /
amc_xsum_round: pure host functions) against the oracle's.  The device side is checked in tests/test_gpu_xsum.py."""
import math

import numpy as np
import pytest

import oracle_lib as O
import xsum_model as M
from montecarlo_amd import _capi as A


def _wild(rng, n):
    """Doubles over many binades, both signs, with zeros, subnormals and exact powers of two mixed in."""
    v = rng.standard_normal(n) * np.exp2(rng.integers(-60, 60, n).astype(np.float64))
    v[rng.integers(0, n, n // 10)] = 0.0
    v[rng.integers(0, n, n // 20)] = -0.0
    v[rng.integers(0, n, n // 20)] = np.exp2(rng.integers(-50, 50, n // 20).astype(np.float64))
    v[rng.integers(0, n, 3)] = 5e-324
    return v


@pytest.mark.parametrize("seed", range(6))
def test_running_top_sum_matches_the_big_integer_model(seed):
    rng = np.random.default_rng(seed)
    v = _wild(rng, 400)
    top, k1, k2, value = M.sum_r(list(v))
    rec = O.xsum_r(v)
    kind, e, flags, r1, r2 = M.record_ints(rec)
    assert (kind, e, flags) == (2, top, 0)
    assert (r1, r2) == (k1, k2)
    assert O.xsum_round(rec)[0] == value
    # close to the exact sum: the worst-case quantum is 2^-49 of the largest magnitude
    exact = math.fsum(v)
    assert abs(value - exact) <= len(v) * 2.0 ** -48 * np.abs(v).max()


@pytest.mark.parametrize("scale", [1e-300, 1e-30, 1.0, 3.0e7, 1e200, 1.0e299])
def test_running_top_sum_at_every_magnitude(scale):
    rng = np.random.default_rng(11)
    v = rng.standard_normal(300) * 0.3 * scale
    v = v[np.isfinite(v)]
    top, k1, k2, value = M.sum_r(list(v))
    rec = O.xsum_r(v)
    assert M.record_ints(rec) == (2, top, 0, k1, k2)
    assert O.xsum_round(rec)[0] == value
    assert value == pytest.approx(math.fsum(v), rel=1e-9, abs=0.0) or abs(math.fsum(v)) < 1e-9 * np.abs(v).sum()


def test_running_top_sum_does_not_depend_on_the_order_or_the_split():
    rng = np.random.default_rng(5)
    # values whose running top rises by one and by several levels, in every order
    v = np.concatenate([rng.standard_normal(200) * 0.4, rng.standard_normal(50) * 1e-9, [0.75, 2.0 ** 49, -2.0 ** 48.5, 3.1e16],
                        rng.standard_normal(5) * 1e40])
    whole = O.xsum_r(v)
    for trial in range(20):
        p = rng.permutation(v)
        assert np.array_equal(O.xsum_r(p), whole)
        cuts = np.sort(rng.integers(0, p.size, 5))
        parts = [O.xsum_r(c) for c in np.split(p, cuts)]
        order = rng.permutation(len(parts))
        acc_o = np.zeros((1, O.XS_WORDS))
        acc_a = np.zeros((1, A.AMC_XSUM_WORDS))
        for i in order:
            acc_o = O.xsum_merge(acc_o, parts[i])          # the oracle's merge
            acc_a = A.xsum_merge(acc_a, parts[i])          # libamc.so's host-side merge
        assert np.array_equal(acc_o[0], whole)
        assert np.array_equal(acc_a[0], whole)
    assert A.xsum_round(whole)[0] == O.xsum_round(whole)[0] == M.sum_r(list(v))[3]


def test_a_summand_just_below_half_a_quantum_of_the_next_level_contributes_the_same_early_and_late():
    # the case that fixes the level rule: |v| < 2^(50 l + 49), so that a value taken at level l rounds to 0 at level l + 1
    big = 2.0 ** 100 * 1.5                     # forces top = 2 (|v| >= 2^99)
    for small in (2.0 ** 48 * 1.999, 2.0 ** 49 * 1.0000001, 2.0 ** 98.9, 0.9, 2.0 ** -1 * 0.99999):
        a = O.xsum_r([small, big])
        b = O.xsum_r([big, small])
        assert np.array_equal(a, b)
        assert O.xsum_round(a)[0] == M.sum_r([small, big])[3]


def test_running_top_sum_beyond_the_last_level():
    v = [1.7e308, 1.7e308, -1.0e300]            # 2^999 = 5.4e300 and more count as infinities
    rec = O.xsum_r(v)
    assert O.xsum_round(rec)[0] == math.inf == A.xsum_round(rec)[0] == M.sum_r(v)[3]
    v = [5.0e300, 5.0e300, -1.0e290, 3.0]       # the last level takes them: the sum is finite
    rec = O.xsum_r(v)
    assert O.xsum_round(rec)[0] == A.xsum_round(rec)[0] == M.sum_r(v)[3] == 1.0e301 - 1.0e290
    assert math.isnan(O.xsum_round(O.xsum_r([6e300, -6e300]))[0])


def test_running_top_sum_flags():
    assert math.isnan(O.xsum_round(O.xsum_r([1.0, math.nan, 2.0]))[0])
    assert O.xsum_round(O.xsum_r([1.0, math.inf, 2.0]))[0] == math.inf
    assert O.xsum_round(O.xsum_r([1.0, -math.inf]))[0] == -math.inf
    assert math.isnan(O.xsum_round(O.xsum_r([math.inf, -math.inf]))[0])
    rec = O.xsum_r([math.inf, 1.0])
    assert A.xsum_round(rec)[0] == math.inf
    assert O.xsum_round(O.xsum_r([]))[0] == 0.0
    assert A.xsum_round(np.zeros(A.AMC_XSUM_WORDS))[0] == 0.0                      # all-zero words: the empty sum
    assert np.array_equal(A.xsum_merge(np.zeros(12), O.xsum_r([1.5]))[0], O.xsum_r([1.5]))


@pytest.mark.parametrize("e", [-60, -41, -34, 0, 7])
def test_fixed_quantum_sum_matches_the_model(e):
    rng = np.random.default_rng(e + 100)
    bound = 2.0 ** (e + 41)
    v = (rng.random(500) * 2 - 1) * bound * 0.999
    v[:20] = (rng.integers(-2 ** 20, 2 ** 20, 20) + 0.5) * 2.0 ** e        # exact half-way points before lsb1
    k, value = M.sum_q(list(v), e)
    rec = O.xsum_q(v, e)
    assert M.record_ints(rec)[:4] == (1, e, 0, k)
    assert O.xsum_round(rec)[0] == value == A.xsum_round(rec)[0]
    for trial in range(5):
        assert np.array_equal(O.xsum_q(rng.permutation(v), e), rec)
    # product columns: the exact product is rounded once
    x = (rng.random(300) * 2 - 1) * 2.0 ** 10
    y = (rng.random(300) * 2 - 1) * 2.0 ** (e + 30)
    k, value = M.sum_q_product(list(x), list(y), e)
    rec = O.xsum_q_product(x, y, e)
    assert M.record_ints(rec)[:4] == (1, e, 0, k)
    assert O.xsum_round(rec)[0] == value


def test_fixed_quantum_sum_nan_and_mismatched_quanta():
    assert math.isnan(O.xsum_round(O.xsum_q([0.5, math.nan], M.E_RATIO))[0])
    assert math.isnan(O.xsum_round(O.xsum_q([0.5, math.inf], M.E_RATIO))[0])
    a, b = O.xsum_q([0.5], -34), O.xsum_q([0.5], -33)
    assert math.isnan(A.xsum_round(A.xsum_merge(a, b))[0])              # records of different quanta do not add
    assert math.isnan(O.xsum_round(O.xsum_merge(a, b))[0])
    assert math.isnan(A.xsum_round(A.xsum_merge(a, O.xsum_r([0.5])))[0])  # nor do records of different kinds


def test_record_rounding_of_wide_integers_is_half_even_at_53_bits():
    # totals far beyond 2^53: the host-side rounding of libamc.so against Python's correctly rounded int -> float
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(2, 40))
        v = (rng.random(n) * 2 - 1) * 2.0 ** 40 * 0.99
        e = -int(rng.integers(0, 12))
        k, value = M.sum_q(list(v) * 997, e)
        rec = O.xsum_q(np.tile(v, 997), e)
        assert A.xsum_round(rec)[0] == value == O.xsum_round(rec)[0]
    # an exact tie at 53 bits goes to even
    rec = O.xsum_q([2.0 ** 53], 0)
    rec = A.xsum_merge(rec, O.xsum_q([1.0], 0))[0]
    assert M.record_ints(rec)[3] == 2 ** 53 + 3                        # lsb1(2^53) = 2^53 + 2
    assert A.xsum_round(rec)[0] == float(2 ** 53 + 3) == 2.0 ** 53 + 4.0   # half way between 2^53 + 2 and 2^53 + 4: to even
    assert A.xsum_round(rec)[0] == O.xsum_round(rec)[0]


def test_gradient_data_quanta_bound_every_summand():
    """xs_gd_exponents: with |z| <= 8.5 no summand of (j, grad j, grad logq, g) reaches 2^(E + 46), so a lane's 2^5 summands
    stay inside the accumulator's binade."""
    rng = np.random.default_rng(9)
    for sigma in [1e-70, 3.3e-7, 0.1, 0.2, 0.5, 1.0, 1.2, 1.9999, 2.0, 77.0, 1e70]:
        e = O.gd_exponents(sigma)
        worst = [0.0] * 4
        for z in list(rng.uniform(-8.5, 8.5, 200)) + [8.5, -8.5, 0.0, 1.0]:
            for x in (0.0, 0.3 * sigma, -2.0 * sigma):
                s, _ = O.pg_summands("harmonic", 1e-9 / (sigma * sigma), sigma, z, x)   # a flat target: alpha ~ 1, the bounds are attained
                for i in range(4):
                    worst[i] = max(worst[i], abs(s[i]))
        for i in range(4):
            assert worst[i] < 2.0 ** (e[i] + 46), (sigma, i, worst[i], e[i])
            assert worst[i] > 2.0 ** (e[i] + 46 - 5), "the bound is within 2^5 of what occurs"


def test_spec_form_summands_are_within_ulps_of_the_reference_ordered_form():
    rng = np.random.default_rng(10)
    for trial in range(2000):
        sigma = float(np.exp(rng.uniform(-3, 1)))
        z = float(rng.standard_normal())
        x = float(rng.standard_normal() * 0.5)
        pot = "harmonic" if trial % 2 else "double_well"
        a, xa = O.pg_summands(pot, 2.0, sigma, z, x, "spec")
        b, xb = O.pg_summands(pot, 2.0, sigma, z, x, "reference")
        assert xa == xb                                            # the state is the reference's, bit for bit
        scale = [abs(b[0]), abs(b[0]) * (z * z + 1) / sigma, (z * z + 1) / sigma, ((z * z + 1) / sigma) ** 2]
        for i in range(4):
            assert abs(a[i] - b[i]) <= 64 * 2.0 ** -52 * scale[i] + 1e-300


def test_oracle_callback_sums_agree_with_the_plain_left_to_right_sums():
    sim = O.OracleSim(3001, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=4)
    sim.init_uniform(-2.0, 2.0)
    sim.make_steps(40)
    assert sim.energy() == pytest.approx(sim.energy_plain(), rel=1e-13)
    assert np.allclose(sim.acceptance(), sim.acceptance_plain(), rtol=1e-10, atol=0.0)
    rec = sim.callback_records()
    assert O.xsum_round(rec)[3] == 3001.0
    assert O.xsum_round(rec)[0] / 3001.0 == sim.energy()
    g, gp = sim.pg_estimate([1], 3), None
    sim2 = O.OracleSim(3001, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=4)
    sim2.init_uniform(-2.0, 2.0)
    sim2.make_steps(40)
    gp = sim2.pg_estimate_plain([1], 3)
    assert np.allclose(g, gp, rtol=1e-10, atol=1e-10)
    assert np.array_equal(sim.state()[0], sim2.state()[0])
