"""Pins the CPU oracle: published KATs of its third-party pieces, the reference's closed forms,
the committed golden vectors, and per-function behaviour of the restated Julia code (CPU only)."""
import json
import math
import os
import random

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fh(lst):
    return np.array([float.fromhex(v) for v in lst])


def ulp_err(a, b):
    return 0.0 if a == b else abs(a - b) / math.ulp(b)


# ---- Philox4x32-10 and the draw schedule ------------------------------------------------------
def test_philox_random123_kat(oracle):
    for kat in load_json("philox_kats.json")["random123_kat"]:
        assert oracle.philox(kat["ctr"], kat["key"]) == kat["out"]


def test_draw_schedule_golden(oracle):
    for d in load_json("philox_kats.json")["draw_schedule"]:
        assert oracle.counter(d["pair"], d["t"], d["draw"], d["stream"]) == d["counter"]
        w = oracle.draw_words(d["seed"], d["pair"], d["t"], d["draw"], d["stream"])
        assert w == d["words"]
        assert list(oracle.box_muller(w)) == list(fh(d["box_muller"]))
        lib = oracle.load()
        assert lib.amo_uniform_co(w[0], w[1]) == float.fromhex(d["u_co"][0])
        assert lib.amo_uniform_co(w[2], w[3]) == float.fromhex(d["u_co"][1])
        assert [list(oracle.spare12(w, h)) for h in (0, 1)] == d["spare12"]
        assert lib.amo_uniform_pick(d["spare12"][0][1], w[0]) == float.fromhex(d["pick"])
        assert lib.amo_uniform_accept(d["spare12"][1][0], w[2], w[3]) == float.fromhex(d["u_accept"])


def test_counter_packing_is_rocrand_layout(oracle):
    # x = step low word, y = step[47:32] | draw<<16 | stream<<28, (z, w) = pair id (rocRAND "subsequence")
    c = oracle.counter((7 << 32) | 9, (0xABCD << 32) | 0x12345678, 0xFFF, 0xF)
    assert c == [0x12345678, 0xABCD | (0xFFF << 16) | (0xF << 28), 9, 7]
    # distinct draws / streams / steps / pairs never collide
    seen = set()
    for pair in (0, 1, 2 ** 32):
        for t in (0, 1, 2 ** 32, 2 ** 47):
            for draw in (0, 1, 2, 4095):
                for stream in (0, 1, 2):
                    seen.add(tuple(oracle.counter(pair, t, draw, stream)))
    assert len(seen) == 3 * 4 * 4 * 3


def test_uniform_maps(oracle):
    """52 random bits -> significand of a double in [1,2) (or [2,4)), Julia's rand(Float64) construction."""
    lib = oracle.load()
    assert lib.amo_uniform_co(0, 0) == 0.0                                   # [0, 1): 0 included ...
    assert lib.amo_uniform_co(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -52    # ... 1 excluded (Julia rand())
    assert lib.amo_uniform_co(0, 1 << 31) == 0.5 and lib.amo_uniform_co(1 << 12, 0) == 2.0 ** -52
    assert lib.amo_uniform_co((1 << 12) - 1, 0) == 0.0                       # low 12 bits of the low word unused
    assert lib.amo_uniform_oc(0, 0) == 1.0 and lib.amo_uniform_oc(0xFFFFFFFF, 0xFFFFFFFF) == 2.0 ** -52   # (0, 1]
    # spec v5.  Box-Muller angle: the top 28 bits of ONE word, 2 - (w >> 4) 2^-27 in (0, 2]
    assert lib.amo_angle28(0) == 2.0 and lib.amo_angle28(0xFFFFFFFF) == 2.0 ** -27 and lib.amo_angle28(0xF) == 2.0
    assert lib.amo_angle28(0x10) == 2.0 - 2.0 ** -27 and lib.amo_angle28(1 << 31) == 1.0
    # spare bits of the normal draw (x[11:0], z, w[3:0]) -> (accept12, pick12) of the even and the odd chain
    assert oracle.spare12([0xABCDE123, 0, 0, 0], 0) == (0x123, 0) and oracle.spare12([0, 0xFFFFFFFF, 0x00000456, 0xFFFFFFF0], 0) == (0, 0x456)
    assert oracle.spare12([0xFFFFFFFF, 0xFFFFFFFF, 0x00789000, 0xFFFFFFF0], 1) == (0x789, 0)
    assert oracle.spare12([0, 0, 0xBC000000, 0x0000000A], 1) == (0, 0xABC) and oracle.spare12([0] * 4, 1) == (0, 0)
    assert oracle.spare12([0xFFFFFFFF] * 4, 0) == (0xFFF, 0xFFF) and oracle.spare12([0xFFFFFFFF] * 4, 1) == (0xFFF, 0xFFF)
    # Move pick: 36 bits, pick12 on top of the low 24 bits of the chain's accept-draw word.  Accept uniform: 52-bit
    # significand whose top 12 bits are accept12 and whose other 40 are the top 40 bits of the accept-draw word.
    assert lib.amo_uniform_pick(0, 0) == 0.0 and lib.amo_uniform_pick(0xFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -36
    assert lib.amo_uniform_pick(0, 0xFF000000) == 0.0 and lib.amo_uniform_pick(0x800, 0) == 0.5 and lib.amo_uniform_pick(0, 1) == 2.0 ** -36
    assert lib.amo_uniform_pick(1, 0) == 2.0 ** -12 and lib.amo_uniform_pick(0xFFFFF001, 0) == 2.0 ** -12      # 12 bits taken
    assert lib.amo_uniform_accept(0, 0, 0) == 0.0 and lib.amo_uniform_accept(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -52
    assert lib.amo_uniform_accept(0x800, 0, 0) == 0.5 and lib.amo_uniform_accept(0xFFFFF001, 0, 0) == 2.0 ** -12
    assert lib.amo_uniform_accept(0, 1 << 24, 0) == 2.0 ** -52 and lib.amo_uniform_accept(0, (1 << 24) - 1, 0) == 0.0   # W's low 24 bits: the pick's
    assert lib.amo_uniform_accept(0, 0, 1 << 31) == 2.0 ** -13
    rnd = random.Random(7)
    for _ in range(2000):
        lo, hi = rnd.getrandbits(32), rnd.getrandbits(32)
        m = ((hi << 32) | lo) >> 12
        assert lib.amo_uniform_co(lo, hi) == m * 2.0 ** -52
        assert lib.amo_uniform_oc(lo, hi) == 1.0 - m * 2.0 ** -52
        a = rnd.getrandbits(32)
        assert lib.amo_uniform_accept(a, lo, hi) == (((a & 0xFFF) << 40) | (((hi << 32) | lo) >> 24)) * 2.0 ** -52
        assert lib.amo_uniform_pick(a, lo) == (((a & 0xFFF) << 24) | (lo & 0xFFFFFF)) * 2.0 ** -36
        assert lib.amo_angle28(hi) == 2.0 - (hi >> 4) * 2.0 ** -27


def test_box_muller_map(oracle):
    """Shape of rocRAND's box_muller_double(uint4) (rocrand_normal.h:78-98): words (x,y) -> u in (0,1],
    (z,w) -> angle in (0,2], z = sqrt(-2 log u) * (sinpi, cospi); checked against libm."""
    lib = oracle.load()
    rnd = random.Random(3)
    for _ in range(3000):
        v = [rnd.getrandbits(32) for _ in range(4)]
        u = lib.amo_uniform_oc(v[0], v[1])
        w = lib.amo_angle28(v[3])
        s = math.sqrt(-2.0 * math.log(u))
        z0, z1 = oracle.box_muller(v)
        r = math.fmod(w, 2.0)
        assert z0 == pytest.approx(s * math.sin(math.pi * r), rel=1e-13, abs=2e-15)
        assert z1 == pytest.approx(s * math.cos(math.pi * r), rel=1e-13, abs=2e-15)
    assert oracle.box_muller([0, 0, 5, 5]) == (0.0, 0.0)               # u = 1: zero radius exactly, never NaN
    z0, z1 = oracle.box_muller([0xFFFFFFFF, 0xFFFFFFFF, 0, 0])         # u = 2^-52: largest radius, angle 2 pi
    assert math.isfinite(z0) and math.isfinite(z1) and math.hypot(z0, z1) == pytest.approx(math.sqrt(2 * 52 * math.log(2)))


def test_normal_moments(oracle):
    z = np.array([oracle.box_muller(oracle.draw_words(11, p, 3, 0, 1)) for p in range(100000)]).ravel()
    assert abs(z.mean()) < 4 / math.sqrt(z.size)
    assert abs(z.var() - 1) < 4 * math.sqrt(2 / z.size)
    assert abs((z ** 4).mean() - 3) < 0.1


# ---- own exp / log / sincospi ------------------------------------------------------------------
def test_exp_accuracy_and_edges(oracle):
    lib = oracle.load()
    rnd = random.Random(1)
    worst = max(ulp_err(lib.amo_exp(x), math.exp(x)) for x in (rnd.uniform(-708, 700) for _ in range(100000)))
    assert worst <= 2.0
    for _ in range(100000):                      # the accept shortcut relies on these two inequalities
        x = rnd.uniform(0, 3)
        assert lib.amo_exp(x) >= 1.0 and lib.amo_exp(-x) <= 1.0
    assert lib.amo_exp(0.0) == 1.0 and lib.amo_exp(-0.0) == 1.0
    assert all(lib.amo_exp(x) >= 1.0 for x in (1e-300, 1e-17, 1e-9, 0.5, 700.0))   # alpha = 1 whenever dlogp >= 0
    assert all(lib.amo_exp(-x) <= 1.0 for x in (1e-300, 1e-17, 1e-9, 0.5, 700.0))
    assert lib.amo_exp(-708.5) == 0.0 and lib.amo_exp(-1e308) == 0.0 and lib.amo_exp(-math.inf) == 0.0
    assert lib.amo_exp(709.5) == math.inf and lib.amo_exp(math.inf) == math.inf
    assert math.isnan(lib.amo_exp(math.nan))
    assert lib.amo_exp(-708.0) > 0.0


def test_log_accuracy_and_edges(oracle):
    lib = oracle.load()
    rnd = random.Random(2)
    xs = [rnd.random() for _ in range(50000)] + [math.exp(rnd.uniform(-700, 700)) for _ in range(50000)]
    assert max(ulp_err(lib.amo_log(x), math.log(x)) for x in xs if x > 0) <= 1.0
    assert lib.amo_log(1.0) == 0.0
    assert lib.amo_log(0.0) == -math.inf and lib.amo_log(math.inf) == math.inf
    assert math.isnan(lib.amo_log(-1.0)) and math.isnan(lib.amo_log(math.nan))
    assert lib.amo_log(5e-324) == pytest.approx(math.log(5e-324), rel=1e-15)
    assert lib.amo_log(2.0 ** -53) == pytest.approx(-53 * math.log(2), rel=1e-15)


def test_logbm_accuracy_sign_and_exact_zero(oracle):
    """Table-driven Box-Muller log: <= 1 ulp, log(1) == 0 exactly, never positive on (0, 1]."""
    lib = oracle.load()
    rnd = random.Random(8)
    us = [lib.amo_uniform_oc(rnd.getrandbits(32), rnd.getrandbits(32)) for _ in range(100000)]
    us += [1.0 - rnd.random() * 0.02 for _ in range(50000)] + [math.exp(rnd.uniform(-36, 0)) for _ in range(50000)]
    us += [1.0, 1.0 - 2.0 ** -52, 1.0 - 2.0 ** -53 * 2, 2.0 ** -52, 0.5, 0.7071067811865476, 0.7071067811865475]
    for u in us:
        got = lib.amo_logbm(u)
        assert got <= 0.0 and ulp_err(got, math.log(u)) <= 1.0, u
    assert lib.amo_logbm(1.0) == 0.0


def test_sincospi_accuracy_and_quadrants(oracle):
    rnd = random.Random(4)
    for _ in range(50000):
        w = oracle.load().amo_angle28(rnd.getrandbits(32))
        s, c = oracle.sincospi(w)
        r = math.fmod(w, 2.0)
        assert abs(s - math.sin(math.pi * r)) < 1e-15 and abs(c - math.cos(math.pi * r)) < 1e-15
        assert abs(s * s + c * c - 1) < 5e-16
    assert oracle.sincospi(0.5) == (1.0, 0.0) and oracle.sincospi(1.0) == (0.0, -1.0)
    assert oracle.sincospi(1.5) == (-1.0, 0.0) and oracle.sincospi(2.0) == (0.0, 1.0)
    assert oracle.sincospi(0.25) == pytest.approx((math.sqrt(0.5), math.sqrt(0.5)), rel=3e-16)


# ---- the particle_1d model (example/particle_1d/particle_1d.jl) -------------------------------------
def test_log_proposal_density_reference_closed_form(oracle):
    """test/ad_backends_test.jl:19-32: delta = 0, sigma = 0.2 -> logq = 0.6904993792294276, dlogq = -5 (1e-10)."""
    k = load_json("reference_kats.json")["ad_backends"]
    lib = oracle.load()
    assert lib.amo_log_proposal_density(k["delta"], k["sigma"]) == pytest.approx(k["logq"], abs=k["atol"])
    assert lib.amo_grad_log_proposal_density(k["delta"], k["sigma"]) == pytest.approx(k["grad_sigma"], abs=k["atol"])
    rnd = random.Random(5)
    for _ in range(2000):
        d, s = rnd.gauss(0, 1), math.exp(rnd.uniform(-3, 2))
        assert lib.amo_log_proposal_density(d, s) == pytest.approx(-d * d / (2 * s * s) - 0.5 * math.log(2 * math.pi * s * s), rel=1e-13, abs=1e-13)
        assert lib.amo_grad_log_proposal_density(d, s) == pytest.approx(d * d / s ** 3 - 1 / s, rel=1e-12, abs=1e-12)
        assert lib.amo_log_proposal_density(d, s) == lib.amo_log_proposal_density(-d, s)   # backward == forward, bit for bit


def test_potentials(oracle):
    lib = oracle.load()
    assert lib.amo_potential(0, 1.5) == 2.25 and lib.amo_potential(1, 1.5) == 1.5625 and lib.amo_potential(1, 1.0) == 0.0


def test_categorical_walk(oracle):
    """Distributions.jl DiscreteNonParametric sampler (metropolis.jl:206): first i with cumsum > draw."""
    lib = oracle.load()
    w = np.array([0.4, 0.1, 0.1, 0.4])
    wp = w.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))
    cum = [0.4, 0.4 + 0.1, 0.4 + 0.1 + 0.1]
    for r, want in [(0.0, 0), (0.3999, 0), (0.4, 1), (cum[1], 2), (0.59, 2), (cum[2], 3), (0.9999999, 3)]:
        assert lib.amo_categorical(wp, 4, r) == want
    one = np.array([1.0])
    assert lib.amo_categorical(one.ctypes.data_as(type(wp)), 1, 0.999) == 0


def test_mc_step_semantics(oracle):
    """metropolis.jl:176-190 on explicit draws: strict alpha > u, reject re-applies the negated action."""
    beta, sigma = 2.0, 0.1
    # downhill move: alpha == 1 -> accepted for every u in [0, 1)
    a, x, e = oracle.mc_step_explicit(0, beta, sigma, -1.0, 1.0 - 2.0 ** -53, 1.0, 1.0)
    assert a == 1 and x == 1.0 + (0.0 + sigma * -1.0) and e == x * x
    # uphill: alpha = exp(-beta*(e2-e1)) evaluated as in the reference
    x0 = 1.0
    d = 0.0 + sigma * 1.0
    x1 = x0 + d
    lq = oracle.load().amo_log_proposal_density(d, sigma)
    alpha = oracle.load().amo_exp(((-(x1 * x1)) * beta - (-(x0 * x0)) * beta + lq) - lq)
    a, x, e = oracle.mc_step_explicit(0, beta, sigma, 1.0, math.nextafter(alpha, 0.0), x0, x0 * x0)
    assert a == 1 and x == x1
    a, x, e = oracle.mc_step_explicit(0, beta, sigma, 1.0, alpha, x0, x0 * x0)       # alpha > u is strict
    assert a == 0 and x == (x0 + d) + (-d) and e == x * x
    # the revert is NOT a restore: (x + d) + (-d) can differ from x in the last bit
    diffs = 0
    rnd = random.Random(6)
    for _ in range(2000):
        xs, z = rnd.uniform(-2, 2), rnd.gauss(0, 1)
        a, x, e = oracle.mc_step_explicit(0, beta, sigma, z, 1.0 - 2.0 ** -53, xs, xs * xs)
        if a == 0:
            dd = 0.0 + sigma * z
            assert x == (xs + dd) + (-dd)
            diffs += x != xs
    assert diffs > 0
    # NaN / inf states: alpha is NaN (Julia min propagates NaN) -> never accepted
    for bad in (math.inf, -math.inf, math.nan):
        a, x, e = oracle.mc_step_explicit(0, beta, sigma, 0.3, 0.0, bad, bad * bad)
        assert a == 0


# ---- golden trajectories (self-generated; pins the oracle against drift) ---------------------------
@pytest.mark.parametrize("idx", [6, 7])
def test_golden_trajectories_of_vector_policies_and_mixed_pools(oracle, idx):
    """Round 4's two golden cases (a two-parameter policy; a pool of three policy / action classes), through the engine-shaped
    wrapper of the oracle: states, counters, callback values and the estimator's rows."""
    case = load_json("oracle_trajectories.json")["cases"][idx]
    sp = case["spec"]
    extra = {k: sp[k] for k in ("proposal", "n_params", "classes", "class_of_move") if k in sp}
    o = oracle.OracleEngine(n_chains=sp["M"], chain_offset=sp["offset"], potential=sp["potential"], beta=sp["beta"], sigma=sp["sigma"],
                            weight=sp["weight"], seed=sp["seed"], sweepstep=sp["sweepstep"], **extra)
    o.init_uniform(-2.0, 2.0)
    done = 0
    for snap in case["snapshots"]:
        o.sweep(snap["sweep"] - done)
        done = snap["sweep"]
        x, e = o.download_state()
        assert np.array_equal(x, fh(snap["x"])) and np.array_equal(e, fh(snap["e"]))
        acc, tot = o.download_counters()
        assert acc.tolist() == snap["accepted"] and tot.tolist() == snap["total"]
        if "energy" in snap:
            red = o.reduce()
            assert red[0] / sp["M"] == float.fromhex(snap["energy"])
            assert np.array_equal(red[4:] / sp["M"], fh(snap["acceptance"]), equal_nan=True)
    o.sweep(256 - done)
    g = o.pg_estimate(list(range(len(sp["sigma"]))), 3)
    assert np.array_equal(g.ravel(), fh(case["pg_estimate_q3"]))
    assert np.array_equal(o.download_state()[0], fh(case["x_after_pg"]))
    oracle.install_vector_policy(1, None)
    oracle.install_policy_classes(None, None)


@pytest.mark.parametrize("idx", range(6))
def test_golden_trajectories(oracle, idx):
    case = load_json("oracle_trajectories.json")["cases"][idx]
    sp = case["spec"]
    o = oracle.OracleSim(sp["M"], chain_offset=sp["offset"], potential=sp["potential"], beta=sp["beta"],
                         sigma=sp["sigma"], weight=sp["weight"], seed=sp["seed"], sweepstep=sp["sweepstep"],
                         dtype=sp.get("dtype", "f64"), scale_expr=sp.get("scale"))
    o.init_uniform(-2.0, 2.0)
    done = 0
    for snap in case["snapshots"]:
        o.make_steps(snap["sweep"] - done)
        done = snap["sweep"]
        x, e = o.state()
        assert np.array_equal(x, fh(snap["x"])) and np.array_equal(e, fh(snap["e"]))
        acc, tot = o.counters()
        assert acc.tolist() == snap["accepted"] and tot.tolist() == snap["total"]
        if "energy" in snap:
            assert o.energy() == float.fromhex(snap["energy"])
            assert np.array_equal(o.acceptance(), fh(snap["acceptance"]), equal_nan=True)
    o.make_steps(256 - done)
    if "pg_estimate_q3" not in case:
        return
    g = o.pg_estimate(list(range(len(sp["sigma"]))), 3)
    assert np.array_equal(g.ravel(), fh(case["pg_estimate_q3"]))
    assert np.array_equal(o.state()[0], fh(case["x_after_pg"]))


def test_threads_do_not_change_results(oracle):
    """collect vs tcollect (metropolis.jl:265): chains are independent, so the result is identical."""
    a = oracle.OracleSim(1001, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=9)
    b = oracle.OracleSim(1001, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=9)
    a.init_uniform(-2, 2)
    b.init_uniform(-2, 2)
    a.make_steps(20, 1)
    b.make_steps(20, 4)
    assert np.array_equal(a.state()[0], b.state()[0]) and np.array_equal(a.counters()[0], b.counters()[0])


def test_shard_invariance_of_the_oracle(oracle):
    """RNG keyed by GLOBAL chain id: any even split of the ensemble reproduces the unsplit run."""
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=5)
    whole = oracle.OracleSim(101, **kw)
    whole.init_uniform(-2, 2)
    whole.make_steps(30)
    xs, accs = [], []
    for start, stop in [(0, 34), (34, 70), (70, 101)]:
        part = oracle.OracleSim(stop - start, chain_offset=start, **kw)
        part.init_uniform(-2, 2)
        part.make_steps(30)
        xs.append(part.state()[0])
        accs.append(part.counters()[0])
    assert np.array_equal(np.concatenate(xs), whole.state()[0])
    assert np.array_equal(np.concatenate(accs, axis=1), whole.counters()[0])


# ---- build_schedule (src/simulation.jl:95-117) ----------------------------------------------------------
def test_build_schedule_reference_values(oracle):
    k = load_json("reference_kats.json")["schedule"]
    s = oracle.build_schedule(k["steps"], k["burn"], k["block"])
    assert len(s) == k["n_entries"] and s[0] == k["first"] and s[-1] == k["last"]
    assert all(b - a == k["stride"] for a, b in zip(s, s[1:]))
    assert oracle.build_schedule(100, 10, 30) == [10, 40, 70, 100]
    assert oracle.build_schedule(100, 10, 45) == [10, 55, 100]          # ∪ [steps]
    assert oracle.build_schedule(10 ** 5, 1000, 10 ** 4) == list(range(1000, 10 ** 5, 10 ** 4)) + [10 ** 5]
    assert oracle.build_schedule(1000, 10, 2.0) == [10, 11, 12, 14, 18, 26, 42, 74, 138, 266, 522, 1000]
    with pytest.raises(ValueError):
        oracle.build_schedule(1000, 10, 1.5)                           # Int(1.5) is an InexactError in Julia


# ---- the reference's call order on the engine's draw schedule (what julia/PhiloxRNG.jl implements) -----------------
class _PhiloxCalls:
    """The per-chain generator of julia/PhiloxRNG.jl restated in Python: the reference's mc_sweep! makes three calls
    per mc_step! -- rand(rng, Categorical(w)), randn via Normal(0, sigma), rand(rng) (metropolis.jl:206,
    particle_1d.jl:57, metropolis.jl:184) -- and the n-th call of chain c maps to (step, kind) = divmod(n, 3):
      kind 0  move pick      12 spare bits of draw 0 on top of the low 24 bits of the chain's word of draw 1
      kind 1  normal         the chain's half of the Box-Muller pair of draw 0
      kind 2  accept uniform 12 (other) spare bits of draw 0 on top of the top 40 bits of the chain's word of draw 1"""

    def __init__(self, oracle, seed, chain):
        self.o, self.lib, self.seed, self.chain, self.calls = oracle, oracle.load(), seed, chain, 0

    def _next(self, want):
        t, kind = divmod(self.calls, 3)
        assert kind in want
        self.calls += 1
        return t, kind

    def rand(self):
        t, kind = self._next((0, 2))
        pair, odd = self.chain >> 1, self.chain & 1
        va = self.o.draw_words(self.seed, pair, t, 1, 1)
        accept12, pick12 = self.o.spare12(self.o.draw_words(self.seed, pair, t, 0, 1), odd)
        if kind == 0:
            return self.lib.amo_uniform_pick(pick12, va[2 * odd])
        return self.lib.amo_uniform_accept(accept12, va[2 * odd], va[2 * odd + 1])

    def randn(self):
        t, _ = self._next((1,))
        return self.o.box_muller(self.o.draw_words(self.seed, self.chain >> 1, t, 0, 1))[self.chain & 1]


def test_reference_call_order_on_the_draw_schedule_reproduces_the_oracle(oracle):
    """mc_sweep! (metropolis.jl:203-212) written against an RNG OBJECT, like the reference, with the generator above:
    positions, energies and per-move counters equal the oracle's counter-indexed sweep bit for bit.  This is the
    mapping the reference's `R=` hook would consume through julia/PhiloxRNG.jl."""
    M, steps, seed, offset = 9, 40, 2 ** 40 + 17, 6
    sigma, weight, beta = [0.3, 1.1, 0.05], [0.5, 0.2, 0.3], 2.0
    o = oracle.OracleSim(M, chain_offset=offset, potential="double_well", beta=beta, sigma=sigma, weight=weight, seed=seed)
    o.init_uniform(-2, 2)
    x0, e0 = o.state()
    o.make_steps(steps)
    xo, eo = o.state()
    acc_o, tot_o = o.counters()
    lib = oracle.load()
    w = (__import__("ctypes").c_double * 3)(*weight)
    for c in range(M):
        rng = _PhiloxCalls(oracle, seed, offset + c)
        x, e = float(x0[c]), float(e0[c])
        acc, tot = [0, 0, 0], [0, 0, 0]
        for _ in range(steps):
            k = lib.amo_categorical(w, 3, rng.rand())                          # :206
            z = rng.randn()                                                    # sample_action!
            u = rng.rand()                                                     # :184 (drawn after the proposal)
            a, x, e = oracle.mc_step_explicit(1, beta, sigma[k], z, u, x, e)   # :176-190
            acc[k] += a
            tot[k] += 1
        assert x == xo[c] and e == eo[c]
        assert acc == acc_o[:, c].tolist() and tot == tot_o[:, c].tolist()


# ---- Float32 state: the oracle's C-float restatement against an independent numpy.float32 one ---------------------
def _np_f32_step(lib, pot, beta, sigma, z, u, x):
    """mc_step! for Particle{Float32} written with numpy.float32 scalars, following Julia's promotion rules:
    delta = Float32(sigma*z); (delta)^2 in Float32, / (2 sigma^2) in Float64; x, e, dlogp in Float32; alpha in Float64."""
    f32 = np.float32

    def potf(v):
        if pot == 1:
            q = f32(f32(v * v) - f32(1))
            return f32(q * q)
        return f32(v * v)

    def logq(d):
        d2 = f32(-f32(d * d))
        return float(d2) / (2.0 * (sigma * sigma)) - lib.amo_log(float.fromhex("0x1.921fb54442d18p+2") * (sigma * sigma)) / 2.0

    delta = f32(0.0 + sigma * z)
    lf = logq(delta)
    e1 = potf(x)
    xn = f32(x + delta)
    e2 = potf(xn)
    dlogp = f32(f32(f32(-e2) * beta) - f32(f32(-e1) * beta))
    nd = f32(-delta)
    lb = logq(nd)
    alpha = min(1.0, lib.amo_exp(float(dlogp) + lb - lf))
    if alpha > u:
        return 1, xn, e2
    xr = f32(xn + nd)
    return 0, xr, potf(xr)


def test_float32_step_follows_julias_promotion_rules(oracle):
    import ctypes as C
    lib = oracle.load()
    rng = np.random.default_rng(0)
    f32 = np.float32
    n_acc = 0
    for _ in range(20000):
        pot = int(rng.integers(0, 2))
        beta = f32(rng.uniform(0.5, 3))
        sigma = float(rng.uniform(0.05, 1.5))
        z, u = float(rng.normal()), float(rng.uniform())
        x = f32(rng.uniform(-2, 2))
        xx, ee = C.c_float(x), C.c_float(lib.amo_potential_f32(pot, C.c_float(x)))
        a = lib.amo_mc_step_explicit_f32(pot, beta, sigma, z, u, C.byref(xx), C.byref(ee))
        a2, x2, e2 = _np_f32_step(lib, pot, beta, sigma, z, u, x)
        assert a == a2 and f32(xx.value) == x2 and f32(ee.value) == e2
        n_acc += a
    assert 5000 < n_acc < 19000          # both branches exercised


def test_float32_simulation_keeps_float32_values_and_the_right_distribution(oracle):
    s = oracle.OracleSim(4000, potential="harmonic", beta=2.0, sigma=[0.8], weight=[1.0], seed=3, dtype="f32")
    s.init_uniform(-2, 2)
    s.make_steps(300, 8)
    x, e = s.state()
    assert np.array_equal(x, x.astype(np.float32).astype(np.float64))
    assert np.array_equal(e, (x.astype(np.float32) ** 2).astype(np.float64))
    assert abs(s.energy() - 0.25) < 0.02             # <x^2> = 1/(2 beta)
    # the Float64 run from the same seeds differs only at Float32 rounding level early on
    d = oracle.OracleSim(4000, potential="harmonic", beta=2.0, sigma=[0.8], weight=[1.0], seed=3)
    d.init_uniform(-2, 2)
    f = oracle.OracleSim(4000, potential="harmonic", beta=2.0, sigma=[0.8], weight=[1.0], seed=3, dtype="f32")
    f.init_uniform(-2, 2)
    d.make_steps(3)
    f.make_steps(3)
    assert np.max(np.abs(d.state()[0] - f.state()[0])) < 1e-5
