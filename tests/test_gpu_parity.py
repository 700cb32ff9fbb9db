"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors.  Bit-exact for chain state, energies, accept counts AND every cross-chain sum: the sums are
reproducible integer sums, compared as records word for word with the
oracle's restatement.  (RED_RTOL is what is left for comparisons with the oracle's PLAIN left-to-right
Float64 sums -- the reference's own `mean`, which is order-unstable under foldxt -- and with golden
files written before round 4.)
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RED_RTOL = 1e-10      # the reproducible sum vs a PLAIN left-to-right Float64 sum of the same summands


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def fh(lst):
    return np.array([float.fromhex(v) for v in lst])


# ---- arithmetic spec primitives on the device -------------------------------------------------------
def test_device_math_bit_exact(gpu, oracle):
    lib = oracle.load()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-708, 5, 200000), rng.uniform(-1e-3, 1e-3, 1000),
                         [0.0, -0.0, 1e-300, -1e-300, -708.0, -708.5, -745.0, 709.0, 709.5, np.inf, -np.inf, np.nan]])
    assert np.array_equal(bits(gpu.selftest_math("exp", xs)), bits([lib.amo_exp(v) for v in xs]))
    xs = np.concatenate([rng.uniform(0, 1, 200000), np.exp(rng.uniform(-700, 700, 50000)),
                         [1.0, 2.0 ** -53, 5e-324, 2.2e-308, 0.0, np.inf, -1.0, np.nan]])
    assert np.array_equal(bits(gpu.selftest_math("log", xs)), bits([lib.amo_log(v) for v in xs]))
    us = np.concatenate([1.0 - rng.integers(0, 2 ** 52, 200000) * 2.0 ** -52, 1.0 - rng.uniform(0, 0.02, 50000),
                         np.exp(rng.uniform(-36, 0, 50000)), [1.0, 1.0 - 2.0 ** -52, 2.0 ** -52, 0.5, 0.7071067811865476]])
    assert np.array_equal(bits(gpu.selftest_math("logbm", us)), bits([lib.amo_logbm(v) for v in us]))
    ws = np.concatenate([2.0 - rng.integers(0, 2 ** 52, 200000) * 2.0 ** -51, [0.25, 0.5, 1.0, 1.5, 2.0, 2.0 ** -51]])
    ref = np.array([oracle.sincospi(v) for v in ws])
    assert np.array_equal(bits(gpu.selftest_math("sinpi", ws)), bits(ref[:, 0]))
    assert np.array_equal(bits(gpu.selftest_math("cospi", ws)), bits(ref[:, 1]))


def test_device_sqrt_and_division_are_ieee(gpu):
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(0, 80, 300000), np.exp(rng.uniform(-700, 700, 100000)), [0.0, np.inf, 4.0]])
    assert np.array_equal(bits(gpu.selftest_math("sqrt", a)), bits(np.sqrt(a)))
    # the Box-Muller radius takes sqrt(-2 log u), u in (0,1] on the 2^-52 grid, by the kernel's own unscaled
    # sequence (amc_math.h sqrt_radius_f64): every value class that can occur, incl. exact 0 (u = 1)
    r = np.concatenate([rng.uniform(0, 80, 2_000_000), np.exp(rng.uniform(np.log(2.0 ** -53), np.log(80), 1_000_000)),
                        -2.0 * np.log(1.0 - rng.integers(0, 2 ** 52, 1_000_000) * 2.0 ** -52),
                        [0.0, 2.0 ** -52, 2.0 ** -51, 72.0873, 1.0, 4.0, np.nextafter(4.0, 0), np.nextafter(4.0, 5)]])
    assert np.array_equal(bits(gpu.selftest_math("sqrt_radius", r)), bits(np.sqrt(r)))
    num = -rng.uniform(0, 30, 400000) ** 2
    den = 2 * np.exp(rng.uniform(-6, 3, 400000)) ** 2
    assert np.array_equal(bits(gpu.selftest_math("div", num, den)), bits(num / den))


def test_log_proposal_density_and_gradient_known_answers(gpu, oracle):
    """test/ad_backends_test.jl:19-32 on the device: delta = 0, sigma = 0.2 -> logq = 0.6904993792294276, d logq / d sigma
    = -5.0 (atol 1e-10, the reference's own tolerance).  logq and d logq / d sigma in the reference's operation order
    (ForwardDiff's dual rules; what the host-side withgrad_log_proposal_density returns): bit for bit against the oracle on
    random arguments.  The derivative as the estimator KERNEL forms it (prepare_params' split coefficient; one fma chain
    d^2 (dden/den^2) - dlhalf instead of two divisions, amc_kernels.h pg_sample): within 4 ulp of the two terms it subtracts."""
    logq = gpu.selftest_math("log_proposal_density", [0.0], [0.2])[0]
    dlogq = gpu.selftest_math("grad_log_proposal_density", [0.0], [0.2])[0]
    dlogq_k = gpu.selftest_math("grad_log_proposal_density_kernel_form", [0.0], [0.2])[0]
    assert abs(logq - 0.6904993792294276) < 1e-10 and abs(dlogq - (-5.0)) < 1e-10 and abs(dlogq_k - (-5.0)) < 1e-10
    rng = np.random.default_rng(8)
    n = 200000
    sig = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), n))
    delta = sig * rng.normal(0, 1, n)
    lib = oracle.load()
    want = np.array([lib.amo_log_proposal_density(d, s) for d, s in zip(delta[:20000], sig[:20000])])
    wantg = np.array([lib.amo_grad_log_proposal_density(d, s) for d, s in zip(delta[:20000], sig[:20000])])
    got = gpu.selftest_math("log_proposal_density", delta, sig)
    gotg = gpu.selftest_math("grad_log_proposal_density", delta, sig)
    gotk = gpu.selftest_math("grad_log_proposal_density_kernel_form", delta, sig)
    assert np.array_equal(bits(got[:20000]), bits(want))
    assert np.array_equal(bits(gotg[:20000]), bits(wantg))
    terms = delta[:20000] ** 2 / sig[:20000] ** 3 + 1 / sig[:20000]
    assert np.all(np.abs(gotk[:20000] - wantg) <= 4 * 2.0 ** -52 * terms)
    # closed forms (particle_1d.jl:53 and its sigma-derivative delta^2/sigma^3 - 1/sigma)
    assert np.allclose(got, -delta ** 2 / (2 * sig ** 2) - np.log(2 * np.pi * sig ** 2) / 2, rtol=1e-12, atol=1e-12)
    assert np.allclose(gotg, delta ** 2 / sig ** 3 - 1 / sig, rtol=1e-10, atol=1e-12)
    assert np.allclose(gotk, delta ** 2 / sig ** 3 - 1 / sig, rtol=1e-10, atol=1e-12)


def test_ad_backends_test_of_the_reference(gpu):
    """test/ad_backends_test.jl, line for line on this package's names: the value and the sigma-gradient of
    log_proposal_density for action = Displacement(0.0), parameters = (sigma = 0.2) -- there the three AD backends must
    agree to 1e-10; here the device's closed form must agree with the value they agree on."""
    import montecarlo_amd as ma
    action = ma.Displacement(0.0)
    policy = ma.StandardGaussian()
    parameters = {"sigma": 0.2}
    grad = np.zeros(1)
    logq = ma.withgrad_log_proposal_density(grad, action, policy, parameters)
    assert logq == pytest.approx(0.6904993792294276, abs=1e-10)
    assert grad[0] == pytest.approx(-5.0, abs=1e-10)
    assert ma.log_proposal_density(action, policy, parameters) == logq
    with pytest.raises(NotImplementedError, match="No log_proposal_density is defined"):
        ma.log_proposal_density(action, ma.ScaledGaussian("1.0 + x*x"), parameters)


def test_reciprocal_correction_division_is_ieee(gpu):
    """The sweep kernel divides by den = 2 sigma^2 with a precomputed reciprocal and two Markstein
    corrections (amc_kernels.h div_by_const); it must equal the IEEE quotient for every input."""
    rng = np.random.default_rng(4)
    n = 2_000_000
    sig = np.exp(rng.uniform(np.log(1e-100), np.log(1e100), n))
    den = 2.0 * (sig * sig)
    z = np.concatenate([rng.normal(0, 1, n - 6), [0.0, -0.0, 7e-16, 8.6, 1.0, 1e-8]])
    num = -((0.0 + sig * z) ** 2)
    got, want = gpu.selftest_math("div_by_const", num, den), num / den
    nz = num != 0
    assert np.array_equal(bits(got[nz]), bits(want[nz]))
    assert np.all(got[~nz] == 0.0)       # a = -0.0 yields +0.0: logq = q - logc is the same either way
    # adversarial divisors: all-ones / sparse significands, powers of two, and numerators next to them
    m = (np.uint64(0x3FF0000000000000) | rng.integers(0, 2 ** 52, n, dtype=np.uint64)).view(np.float64)
    den = np.concatenate([np.nextafter(2.0, 0) * np.ones(n // 4), np.nextafter(1.0, 2) * np.ones(n // 4),
                          m[: n // 4], 2.0 ** rng.integers(-300, 300, n // 4)])
    num = -m * np.exp(rng.uniform(-60, 5, n))
    assert np.array_equal(bits(gpu.selftest_math("div_by_const", num, den)), bits(num / den))
    q = -m[: n // 2]                       # numerators of the form RN(q*b) and their neighbours: near-exact quotients
    b = m[n // 2:]
    for a in (q * b, np.nextafter(q * b, 0), np.nextafter(q * b, -np.inf)):
        assert np.array_equal(bits(gpu.selftest_math("div_by_const", a, b)), bits(a / b))


def test_device_philox_words(gpu, oracle):
    rng = np.random.default_rng(2)
    for seed, draw, stream in [(1, 0, 1), (0x123456789ABCDEF, 5, 2), (2 ** 64 - 1, 4095, 0)]:
        pairs = rng.integers(0, 2 ** 40, 500).astype(np.uint64)
        ts = rng.integers(0, 2 ** 47, 500).astype(np.uint64)
        dev = gpu.selftest_philox(seed, pairs, ts, draw, stream)
        ref = np.array([oracle.draw_words(seed, int(p), int(t), draw, stream) for p, t in zip(pairs, ts)], dtype=np.uint32)
        assert np.array_equal(dev, ref)
    kat = json.load(open(os.path.join(GOLDEN, "philox_kats.json")))
    for d in kat["draw_schedule"]:
        dev = gpu.selftest_philox(d["seed"], np.array([d["pair"]], dtype=np.uint64), np.array([d["t"]], dtype=np.uint64),
                                  d["draw"], d["stream"])
        assert dev[0].tolist() == d["words"]


# ---- the sweep: HIP vs oracle ----------------------------------------------------------------------------
POOLS = {1: ([0.1], [1.0]), 2: ([0.1, 1.0], [0.5, 0.5]), 3: ([0.1, 1.0, 0.3], [0.2, 0.5, 0.3]),
         5: ([0.1, 1.0, 0.3, 0.5, 0.2], [0.2, 0.3, 0.1, 0.25, 0.15]), 7: ([0.2] * 7, [0.4] + [0.1] * 6),
         8: ([0.1 * (k + 1) for k in range(8)], [0.125] * 8),
         # more than eight moves: one byte per chain in the step log and the generic (one chain per thread) fold
         11: ([0.1 + 0.05 * k for k in range(11)], [0.2] + [0.08] * 10),
         33: ([0.05 * (k + 1) for k in range(33)], [1.0 / 33] * 33), 64: ([0.02 * (k + 1) for k in range(64)], [1.0 / 64] * 64)}


def run_pair(gpu, oracle, M, K, potential, sweeps, sweepstep=1, counters=True, offset=0, fused=False,
             beta=2.0, beta_arr=None, seed=7, x0=None):
    sigma, weight = POOLS[K]
    e = gpu.HipEngine(n_chains=M, chain_offset=offset, n_chains_global=offset + M, potential=potential, beta=beta,
                      sigma=sigma, weight=weight, seed=seed, sweepstep=sweepstep, per_chain_counters=counters)
    o = oracle.OracleSim(M, chain_offset=offset, potential=potential, beta=beta, sigma=sigma, weight=weight, seed=seed,
                         sweepstep=sweepstep)
    if x0 is None:
        e.init_uniform(-2, 2)
        o.init_uniform(-2, 2)
    else:
        e.upload_state(x0)
        o.set_x(x0)
    if beta_arr is not None:
        e.upload_state(o.state()[0], beta_arr)
        o.set_beta(beta_arr)
    if fused:
        e.sweep(sweeps)
    else:
        for _ in range(sweeps):
            e.sweep(1)
    o.make_steps(sweeps)
    return e, o


def assert_same(e, o, counters=True):
    x, en = e.download_state()
    xo, eo = o.state()
    assert np.array_equal(bits(x), bits(xo)), f"x differs on {np.count_nonzero(bits(x) != bits(xo))} chains"
    assert np.array_equal(bits(en), bits(eo))
    acc, tot = e.counter_totals()
    ao, to = o.counters()
    assert np.array_equal(acc, ao.sum(1)) and np.array_equal(tot, to.sum(1))      # accept counts bit-exact
    if counters:
        a2, t2 = e.download_counters()
        assert np.array_equal(a2, ao) and np.array_equal(t2, to)
    # the callback sums: the records of the device's integer sums are the oracle's, word for word (a K = 1 engine without
    # per-chain counters carries the pool-wide accepted total in the ratio record: amc_reduce_end_exact)
    from oracle_lib import xsum_q
    rec, steps = e.reduce_exact()
    want = o.callback_records()
    if e.n_moves == 1 and not e.per_chain_counters:
        want[4] = xsum_q([float(ao.sum())], 0)
    assert np.array_equal(rec, want), [i for i in range(rec.shape[0]) if not np.array_equal(rec[i], want[i])]
    red = e.reduce_records_value(rec, steps)
    M = x.size
    assert red[3] == M
    # (NaN where a chain never picked a move, or met a NaN: equal as NaN, whatever its sign bit)
    assert np.array_equal(red[0] / M, o.energy(), equal_nan=True)
    assert np.array_equal(red[1:3], o.moments(), equal_nan=True)
    if e.n_moves == 1 and not e.per_chain_counters:
        # (sum_c accepted_c) / total against sum_c (accepted_c / total): one rounding against M of them
        np.testing.assert_allclose(red[4:] / M, o.acceptance(), rtol=1e-13, equal_nan=True)
    else:
        assert np.array_equal(red[4:] / M, o.acceptance(), equal_nan=True)


@pytest.mark.parametrize("M,K,potential,sweeps,sweepstep", [
    (1, 1, "harmonic", 10, 1),                # a single chain (half-empty pair)
    (2, 2, "double_well", 10, 1),
    (10, 1, "harmonic", 200, 1),              # BASELINE config 1 shape
    (1000, 1, "harmonic", 50, 1),
    (1001, 1, "harmonic", 20, 3),             # odd M: ragged tail lane
    (4097, 2, "double_well", 30, 2),          # BASELINE config 3 shape
    (777, 3, "harmonic", 25, 1),
    (513, 7, "harmonic", 12, 1),              # pgmc_test.jl pool: 7 moves
    (65536, 1, "harmonic", 8, 1),
])
def test_sweep_bit_exact(gpu, oracle, M, K, potential, sweeps, sweepstep):
    e, o = run_pair(gpu, oracle, M, K, potential, sweeps, sweepstep)
    assert_same(e, o)
    e.close()


@pytest.mark.parametrize("depth", [1, 3, 255])
@pytest.mark.parametrize("K,fused", [(1, False), (2, False), (2, True), (5, False), (7, True), (8, True), (11, True), (33, False), (64, True)])
def test_step_log_depths(gpu, oracle, monkeypatch, depth, K, fused):
    """Per-chain counters go through the step log (a nibble per chain and MH step up to four moves, a byte beyond) and are
    folded into acc/tot on demand or when the log is full: every depth, single-step and multi-step launches, register and
    generic fold."""
    monkeypatch.setenv("AMC_LOG_DEPTH", str(depth))
    e, o = run_pair(gpu, oracle, 1001, K, "double_well", 11, sweepstep=2, fused=fused)
    assert_same(e, o)
    e.sweep(5)                    # counters were folded by the download above; the log starts over
    o.make_steps(5)
    assert_same(e, o)
    e.close()


def test_wide_reduction_rows_use_device_final_passes(gpu, oracle):
    """K = 7 gives 11 columns per partial row (> 8: the host-summed 64-byte rows do not apply) and 200 001 chains
    give 782 rows (> 256): the two-level device final pass of amc_reduce."""
    e, o = run_pair(gpu, oracle, 200_001, 7, "double_well", 4)
    assert_same(e, o)
    e.close()


def test_sweep_bit_exact_beyond_one_grid(gpu, oracle):
    """More pairs than 2048 x 256 threads: the grid-stride path, sweeps x M = 3.6e6 updates."""
    e, o = run_pair(gpu, oracle, 1_200_001, 1, "harmonic", 3)
    assert_same(e, o)
    e.close()


@pytest.mark.parametrize("counters", [False, True])
def test_sweep_bit_exact_three_grid_stride_iterations(gpu, oracle, counters):
    """3e6 chains = 1.5e6 pairs = three grid-stride iterations per thread (full, full, ragged): exercises the
    prefetch / delayed-store schedule of both sweep forms, two sweeps so stored values are re-read."""
    e = gpu.HipEngine(n_chains=3_000_001, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=7,
                      per_chain_counters=counters)
    o = oracle.OracleSim(3_000_001, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=7)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    e.sweep(1)
    e.sweep(1)
    o.make_steps(2, oracle.load().amo_max_threads())
    assert_same(e, o, counters=counters)
    e.close()


def test_pool_wide_counter_mode(gpu, oracle):
    """per_chain_counters = 0 (the benchmark's 16 B/update mode): x and the accepted total still exact."""
    e, o = run_pair(gpu, oracle, 5000, 1, "harmonic", 20, counters=False)
    assert_same(e, o, counters=False)
    with pytest.raises(gpu.AmcError, match="per_chain_counters"):
        e.download_counters()
    e.close()


def test_fused_launch_equals_separate_launches(gpu, oracle):
    e, o = run_pair(gpu, oracle, 3000, 2, "harmonic", 16, sweepstep=2, fused=True)
    assert_same(e, o)
    e.close()


def test_shard_at_offset_and_shard_invariance_on_one_device(gpu, oracle):
    e, o = run_pair(gpu, oracle, 3001, 2, "harmonic", 16, offset=123456)
    assert_same(e, o)
    e.close()
    # the same global range as 1 shard and as 3 shards on one device
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=5)
    whole = gpu.HipEngine(n_chains=1001, **kw)
    whole.init_uniform(-2, 2)
    whole.sweep(25)
    parts = []
    for start, stop in [(0, 334), (334, 700), (700, 1001)]:
        p = gpu.HipEngine(n_chains=stop - start, chain_offset=start, n_chains_global=1001, **kw)
        p.init_uniform(-2, 2)
        p.sweep(25)
        parts.append((p.download_state()[0], p.download_counters()[0]))
        p.close()
    assert np.array_equal(bits(np.concatenate([p[0] for p in parts])), bits(whole.download_state()[0]))
    assert np.array_equal(np.concatenate([p[1] for p in parts], axis=1), whole.download_counters()[0])
    whole.close()


def test_per_chain_beta(gpu, oracle):
    b = np.random.default_rng(1).uniform(1.0, 3.0, 2049)
    e, o = run_pair(gpu, oracle, 2049, 1, "double_well", 16, beta_arr=b)
    assert_same(e, o)
    e.close()


def test_uploaded_state_and_nonfinite_chains(gpu, oracle):
    """Edge cases the domain has: chains at +-inf / NaN never accept and stay put (alpha = NaN, Julia min);
    huge |x| overflows e to inf; -0.0 survives."""
    x0 = np.random.default_rng(3).uniform(-2, 2, 64)
    x0[[0, 1, 2, 3, 4, 5]] = [np.inf, -np.inf, np.nan, 1e200, -0.0, 1e-320]
    e, o = run_pair(gpu, oracle, 64, 1, "harmonic", 20, x0=x0)
    assert_same(e, o)
    x, _ = e.download_state()
    acc, _ = e.download_counters()
    assert np.isposinf(x[0]) and np.isneginf(x[1]) and np.isnan(x[2]) and np.all(acc[0, :3] == 0)
    e.close()


def test_parameter_update_and_step_counter(gpu, oracle):
    """amc_set_parameters (learning_step! result) and amc_set_step (resume) keep parity."""
    e, o = run_pair(gpu, oracle, 999, 2, "harmonic", 5)
    e.set_parameters(1, [0.37])
    o.set_sigma(1, 0.37)
    assert e.get_parameters(1)[0] == 0.37
    e.sweep(5)
    o.make_steps(5)
    assert_same(e, o)
    assert e.step == 10 == o.step
    e.step = 1 << 33
    o.step = 1 << 33
    e.sweep(3)
    o.make_steps(3)
    x, _ = e.download_state()
    assert np.array_equal(bits(x), bits(o.state()[0]))
    with pytest.raises(gpu.AmcError):
        e.set_parameters(0, [-1.0])
    with pytest.raises(gpu.AmcError):
        e.set_parameters(5, [0.1])
    e.close()


def test_first_row_acceptance_is_nan(gpu):
    """callback_acceptance at t = 0 is 0/0 = NaN per move (store_first row, algorithms.jl:93)."""
    for K, counters in [(1, True), (1, False), (2, True)]:
        sigma, weight = POOLS[K]
        e = gpu.HipEngine(n_chains=100, sigma=sigma, weight=weight, per_chain_counters=counters)
        assert np.all(np.isnan(e.reduce()[4:]))
        e.close()


# ---- golden vectors -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("idx", range(8))
def test_hip_matches_golden_trajectories(gpu, idx):
    case = json.load(open(os.path.join(GOLDEN, "oracle_trajectories.json")))["cases"][idx]
    sp = case["spec"]
    extra = {k: sp[k] for k in ("proposal", "n_params", "classes", "class_of_move") if k in sp}     # cases 6, 7 (round 4)
    e = gpu.HipEngine(n_chains=sp["M"], chain_offset=sp["offset"], n_chains_global=sp["offset"] + sp["M"],
                      potential=sp["potential"], beta=sp["beta"], sigma=sp["sigma"], weight=sp["weight"],
                      seed=sp["seed"], sweepstep=sp["sweepstep"], dtype=sp.get("dtype", "f64"), scale_expr=sp.get("scale"), **extra)
    e.init_uniform(-2.0, 2.0)
    done = 0
    for snap in case["snapshots"]:
        e.sweep(snap["sweep"] - done)
        done = snap["sweep"]
        x, en = e.download_state()
        assert np.array_equal(bits(x), bits(fh(snap["x"]))) and np.array_equal(bits(en), bits(fh(snap["e"])))
        acc, tot = e.download_counters()
        assert acc.tolist() == snap["accepted"] and tot.tolist() == snap["total"]
        if "energy" in snap:
            red = e.reduce()                         # reproducible sums: the oracle's bits, whatever the grid
            assert red[0] / sp["M"] == float.fromhex(snap["energy"])
            assert np.array_equal(red[4:] / sp["M"], fh(snap["acceptance"]), equal_nan=True)
    e.sweep(256 - done)
    if "pg_estimate_q3" not in case:
        e.close()
        return
    g = e.pg_estimate(list(range(len(sp["sigma"]))), 3)
    assert np.array_equal(bits(g.ravel()), bits(fh(case["pg_estimate_q3"])))
    assert np.array_equal(bits(e.download_state()[0]), bits(fh(case["x_after_pg"])))
    e.close()


# ---- policy-gradient estimator --------------------------------------------------------------------------------

def test_pg_sample_summands_within_ulps(gpu, oracle):
    """ONE sample per launch (one chain, one learnable move, q_batch 1): the four GradientData summands the kernel forms
    with its shortened arithmetic (amc_kernels.h pg_sample: alpha = min(1, exp(dlogp)) without the detour through logq,
    d logq / d sigma as one fma chain) against the oracle's reference-ordered values (gradients.jl:93-109), and the
    chain's position bit for bit."""
    rng = np.random.default_rng(21)
    eps = 2.0 ** -52
    worst = np.zeros(4)
    for trial in range(120):
        sigma = float(np.exp(rng.uniform(np.log(0.02), np.log(3.0))))
        beta = float(rng.uniform(0.5, 4.0))
        pot = ["harmonic", "double_well"][trial % 2]
        kw = dict(potential=pot, beta=beta, sigma=[sigma], weight=[1.0], seed=int(rng.integers(1, 2 ** 40)))
        x0 = np.array([rng.normal(0, 1)])
        e = gpu.HipEngine(n_chains=1, **kw)
        o = oracle.OracleSim(1, **kw)
        e.upload_state(x0)
        o.set_x(x0)
        for call in range(4):
            got, want = e.pg_estimate([0], 1)[0], o.pg_estimate([0], 1)[0]
            assert got[4] == want[4] == 1
            j, dj, dq, g = want[:4]
            # alpha moves by <= 2^-53 (2|dlogp| + |logq|) relative; dlogq by <= 4 ulp of its two terms (d^2/s^3 and 1/s)
            terms = abs(dq) + 2 / sigma
            tol = np.array([64 * eps * abs(j), 64 * eps * abs(j) * terms + 8 * eps * abs(j) * terms, 4 * eps * terms,
                            16 * eps * terms ** 2])
            assert np.all(np.abs(got[:4] - want[:4]) <= tol + 1e-300), (trial, call, got, want)
            worst = np.maximum(worst, np.abs(got[:4] - want[:4]) / (tol + 1e-300))
            assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))
        e.close()
        o.close()
    assert np.all(worst <= 1.0)

@pytest.mark.parametrize("M,learn,q", [(5001, [1, 2], 3), (1000, [0], 1), (64, [0, 1, 2], 10), (2047, [2, 0], 2)])
def test_pg_estimate_parity(gpu, oracle, M, learn, q):
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2, 0.1, 0.4], weight=[0.5, 0.25, 0.25], seed=3)
    e = gpu.HipEngine(n_chains=M, **kw)
    o = oracle.OracleSim(M, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    e.sweep(5)
    o.make_steps(5)
    for _ in range(3):          # the estimator call index advances the Philox counter
        g, go = e.pg_estimate(learn, q), o.pg_estimate(learn, q)
        scale = np.abs(go).max(axis=0) + 1.0
        assert np.all(np.abs(g - go) <= 1e-10 * scale * np.sqrt(M * q))
        assert np.array_equal(g[:, 4], go[:, 4])
        assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))   # (x+d)-d drift reproduced exactly
    e.close()


def test_pg_estimate_extreme_arguments(gpu, oracle):
    """alpha = min(1, exp(arg)) at the ends of its domain (gradients.jl:100-104 with Julia's min): wide moves from large |x|
    send arg far below -708 (alpha = 0 exactly in the arithmetic spec), arg > 0 gives exactly 1, and chains at NaN / +-inf
    propagate NaN into the sums like the reference does.  The kernel forms exp(min(arg, 0)) and repairs the two rare cases
    inside a wave-uniform branch; the oracle keeps the three-way case distinction."""
    M = 4099
    kw = dict(potential="harmonic", beta=2.0, sigma=[12.0, 0.3], weight=[0.5, 0.5], seed=17)
    e, o = gpu.HipEngine(n_chains=M, **kw), oracle.OracleSim(M, **kw)
    x0 = np.linspace(-40.0, 40.0, M)
    x0[[7, 100, 2000]] = [0.0, -0.0, 1e-160]
    e.upload_state(x0)
    o.set_x(x0)
    for _ in range(3):
        g, go = e.pg_estimate([0, 1], 4), o.pg_estimate([0, 1], 4)
        scale = np.abs(go).max(axis=0) + 1.0
        assert np.all(np.isfinite(go)) and np.all(np.abs(g - go) <= 1e-10 * scale * np.sqrt(M * 4))
        assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))
    # a third of the wide move's samples must have hit the alpha == 0 arm: |delta| ~ 12, e grows by > 354 easily
    x = o.state()[0]
    assert np.mean(np.abs(x) > 20) > 0.3
    # non-finite states: the reward sums that touch them are NaN on both sides, the other chains' positions still agree bit for bit
    x1 = x.copy()
    x1[[1, 64, 65, 4098]] = [np.nan, np.inf, -np.inf, 1e200]
    e.upload_state(x1)
    o.set_x(x1)
    g, go = e.pg_estimate([0, 1], 2), o.pg_estimate([0, 1], 2)
    assert np.array_equal(np.isnan(g), np.isnan(go)) and np.isnan(go[:, :2]).all()          # j and grad j; the policy's own sums stay finite
    np.testing.assert_allclose(g[:, 2:], go[:, 2:], rtol=1e-10)
    assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))
    e.close()


def test_pg_estimate_seven_move_pool(gpu, oracle):
    """pgmc_test.jl shape: 7 moves, 6 learnable, q_batch_size 10."""
    sigma, weight = POOLS[7]
    kw = dict(potential="harmonic", beta=2.0, sigma=sigma, weight=weight, seed=42)
    e = gpu.HipEngine(n_chains=10, **kw)
    o = oracle.OracleSim(10, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    g, go = e.pg_estimate([1, 2, 3, 4, 5, 6], 10), o.pg_estimate([1, 2, 3, 4, 5, 6], 10)
    np.testing.assert_allclose(g, go, rtol=1e-11, atol=1e-11)
    assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))
    with pytest.raises(gpu.AmcError):
        e.pg_estimate([0, 1, 2, 3, 4, 5, 6, 0, 1], 1)          # > AMC_MAX_LEARN
    with pytest.raises(gpu.AmcError):
        e.pg_estimate([9], 1)
    e.close()


# ---- histogram / strided snapshot / exact resume (device-side stand-ins for the per-chain text I/O) ----------------
def test_histogram_and_strided_snapshot(gpu):
    M = 100_003
    e = gpu.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=21)
    e.init_uniform(-2, 2)
    e.sweep(30)
    x, _ = e.download_state()
    for lo, hi, nb in [(-2.0, 2.0, 200), (-0.5, 1.25, 7), (-3.0, 3.0, 8192)]:
        got = e.histogram(lo, hi, nb)
        inv_w = nb / (hi - lo)
        inside = (x >= lo) & (x < hi)
        want = np.bincount(np.minimum(((x[inside] - lo) * inv_w).astype(np.int64), nb - 1), minlength=nb)
        assert np.array_equal(got[:nb], want.astype(np.uint64))
        assert (got[nb], got[nb + 1], got[nb + 2]) == ((x < lo).sum(), (x >= hi).sum(), 0) and got.sum() == M
    xn = x.copy()
    xn[[5, 77]] = np.nan
    xn[9] = np.inf
    e.upload_state(xn)
    got = e.histogram(-2.0, 2.0, 10)
    assert got[12] == 2 and got[11] >= 1 and got.sum() == M            # NaN bucket, +inf counted as >= hi
    e.upload_state(x)
    for first, stride, count in [(0, 1, 10), (3, 1000, 100), (M - 1, 1, 1), (1, 2, M // 2)]:
        assert np.array_equal(bits(e.download_strided(first, stride, count)), bits(x[first:first + count * stride:stride][:count]))
    with pytest.raises(gpu.AmcError):
        e.download_strided(0, 1000, 200)                                 # leaves the shard
    with pytest.raises(gpu.AmcError):
        e.histogram(1.0, 1.0, 10)
    e.close()


@pytest.mark.parametrize("K,counters", [(1, True), (1, False), (2, True)])
def test_exact_resume(gpu, K, counters):
    """x + counters + sigma + (seed, step) restore a run bit for bit: the generator is counter-based."""
    sigma, weight = POOLS[K]
    kw = dict(n_chains=5001, potential="harmonic", beta=2.0, sigma=sigma, weight=weight, seed=77, per_chain_counters=counters)
    a = gpu.HipEngine(**kw)
    a.init_uniform(-2, 2)
    a.sweep(40)
    a.pg_estimate([0], 2)
    a.set_parameters(0, [0.17])
    x, _ = a.download_state()
    acc_t, tot_t = a.counter_totals()
    b = gpu.HipEngine(**kw)
    b.upload_state(x)
    b.set_parameters(0, [0.17])
    b.step, b.estimator_step = a.step, a.estimator_step
    if counters or K > 1:
        acc, tot = a.download_counters()
        b.upload_counters(acc, tot)
    else:
        b.set_counter_totals(int(acc_t[0]), int(tot_t[0]) // 5001)
    for eng in (a, b):
        eng.sweep(25)
        eng.pg_estimate([0], 2)
    assert np.array_equal(bits(a.download_state()[0]), bits(b.download_state()[0]))
    assert all(np.array_equal(u, v) for u, v in zip(a.counter_totals(), b.counter_totals()))
    np.testing.assert_array_equal(a.reduce(), b.reduce())
    if counters or K > 1:
        assert all(np.array_equal(u, v) for u, v in zip(a.download_counters(), b.download_counters()))
    a.close()
    b.close()


def test_resume_with_pending_gradient_data(gpu):
    """gradients_data (estimator.jl:84,130) is part of the state when PolicyGradientUpdate is scheduled less often than
    the estimator: amc_pg_set_accumulated restores the device-resident running sums, so the first update after a resume
    averages the same samples as the uninterrupted run."""
    kw = dict(n_chains=6001, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=19)
    a = gpu.HipEngine(**kw)
    a.init_uniform(-2, 2)
    for _ in range(3):
        a.sweep(1)
        a.pg_accumulate([1], 2)
    pending = a.pg_get_accumulated([1])
    assert pending[0, 4] == 3 * 6001 * 2
    x, _ = a.download_state()
    acc, tot = a.download_counters()
    b = gpu.HipEngine(**kw)
    b.upload_state(x)
    b.upload_counters(acc, tot)
    b.step, b.estimator_step = a.step, a.estimator_step
    b.pg_set_accumulated([1], pending)
    assert np.array_equal(bits(b.pg_get_accumulated([1])), bits(pending))
    c = gpu.HipEngine(**kw)                       # the same resume WITHOUT the accumulators: must differ
    c.upload_state(x)
    c.upload_counters(acc, tot)
    c.step, c.estimator_step = a.step, a.estimator_step
    for eng in (a, b, c):
        eng.sweep(1)
        eng.pg_accumulate([1], 2)
        eng.pg_update([1], [1], [0.05], [0.0])    # VPG
        eng.sweep(2)
    assert a.get_parameters(1)[0] == b.get_parameters(1)[0] != 0.1
    assert a.get_parameters(1)[0] != c.get_parameters(1)[0]
    assert np.array_equal(bits(a.download_state()[0]), bits(b.download_state()[0]))
    with pytest.raises(gpu.AmcError):
        b.pg_set_accumulated([1], np.array([[0.0, 0.0, 0.0, 0.0, 1.5]]))      # n is a sample count
    with pytest.raises(gpu.AmcError):
        b.pg_set_accumulated([2], pending)
    for eng in (a, b, c):
        eng.close()


def test_estimator_between_reduce_begin_and_end(gpu, oracle):
    """The asynchronous reduction advertised by the C ABI (amc_reduce_begin ... amc_reduce_end) keeps its result while an
    estimator call runs in between -- for wide rows (K > 4) both used to land in one pinned buffer."""
    K = 6
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.1 * (k + 1) for k in range(K)], weight=[0.25, 0.15, 0.15, 0.15, 0.15, 0.15], seed=5)
    e = gpu.HipEngine(n_chains=9001, **kw)
    o = oracle.OracleSim(9001, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    e.sweep(30)
    o.make_steps(30)
    want = e.reduce()
    e.reduce_begin()
    g = e.pg_estimate([0, 1, 2, 3, 4, 5], 1)
    got = e.reduce_end()
    np.testing.assert_array_equal(got, want)
    assert abs(got[0] / 9001 - o.energy()) < 1e-10
    np.testing.assert_allclose(g, o.pg_estimate([0, 1, 2, 3, 4, 5], 1), rtol=1e-10, atol=1e-10)
    e.close()


def test_rccl_through_the_c_abi_single_rank(gpu, oracle):
    """amc_comm_unique_id / amc_comm_init / amc_allreduce_sum (RCCL via dlopen, what the Julia binding uses) and
    the in-place all-reduce inside amc_pg_accumulate, on a 1-rank communicator: values must be unchanged."""
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=3)
    e = gpu.HipEngine(n_chains=20001, **kw)
    o = oracle.OracleSim(20001, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    e.sweep(3)
    o.make_steps(3)
    uid = gpu.HipEngine.comm_unique_id()
    assert len(uid) == 128
    assert e.comm_info() == {"n_ranks": 1, "rank": 0, "rccl_version": 0, "librccl": ""}      # no communicator yet
    e.comm_init(0, 1, uid)
    info = e.comm_info()                         # what RCCL itself reports (ncclCommCount / ncclCommUserRank / ncclGetVersion)
    assert info["n_ranks"] == 1 and info["rank"] == 0 and info["rccl_version"] > 20000 and "librccl" in info["librccl"]
    with pytest.raises(gpu.AmcError, match="already has a communicator"):
        e.comm_init(0, 1, uid)
    a = np.array([1.5, -2.0, 3.25, 1e300])
    assert np.array_equal(e.allreduce_sum(a), a)
    e.pg_accumulate([1], 2)
    acc = e.pg_get_accumulated([1])
    want = o.pg_estimate([1], 2)
    np.testing.assert_allclose(acc, want, rtol=1e-10)
    assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))
    # one communicator, two streams: host-side sums (communication stream) between estimator all-reduces (engine stream)
    for i in range(20):
        e.pgmc_steps(3, [1], 1, [1], [0.01], [0.0])
        assert np.array_equal(e.allreduce_sum(a + i), a + i)
    # back to a single shard, and connected again
    e.comm_destroy()
    assert e.comm_info()["librccl"] == "" and np.array_equal(e.allreduce_sum(a), a)
    e.comm_init(0, 1, gpu.HipEngine.comm_unique_id())
    assert e.comm_info()["n_ranks"] == 1 and np.array_equal(e.allreduce_sum(a), a)
    rt = gpu.runtime_info()
    assert rt["hip_runtime_version"] > 60000000 and "libamdhip64" in rt["hip_runtime"]
    e.close()


def test_extreme_ids_seeds_and_step_indices(gpu, oracle):
    """64-bit seeds, chain offsets beyond 2^32 pairs and step indices beyond 2^32 all reach the Philox counter
    unchanged (DESIGN.md section 3.1 packing)."""
    sigma, weight = POOLS[2]
    for seed, offset, step in [(2 ** 64 - 1, 2 ** 34, 2 ** 32 - 3), (0x0123456789ABCDEF, 2 ** 40 + 2, 2 ** 47 - 100)]:
        kw = dict(potential="harmonic", beta=2.0, sigma=sigma, weight=weight, seed=seed)
        e = gpu.HipEngine(n_chains=3001, chain_offset=offset, n_chains_global=offset + 3001, **kw)
        o = oracle.OracleSim(3001, chain_offset=offset, **kw)
        e.init_uniform(-2, 2)
        o.init_uniform(-2, 2)
        e.step = step
        o.step = step
        e.sweep(7)                       # crosses 2^32 in the first case
        o.make_steps(7)
        assert_same(e, o)
        e.close()
    e = gpu.HipEngine(n_chains=10, sigma=[0.1], weight=[1.0])
    with pytest.raises(gpu.AmcError):
        e.step = 2 ** 48                 # 48-bit step index
    e.close()


def test_largest_pool_and_long_sweepstep(gpu, oracle):
    """AMC_MAX_MOVES = 64 moves with unequal weights; sweepstep = 50 fused steps per make_step!."""
    K = 64
    w = np.arange(1, K + 1, dtype=np.float64)
    w /= w.sum()
    sig = list(np.linspace(0.05, 1.5, K))
    kw = dict(potential="double_well", beta=1.5, sigma=sig, weight=list(w), seed=99, sweepstep=50)
    e = gpu.HipEngine(n_chains=2049, **kw)
    o = oracle.OracleSim(2049, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    e.sweep(4)
    o.make_steps(4, 4)
    assert_same(e, o)
    assert e.step == 200
    with pytest.raises(gpu.AmcError):
        gpu.HipEngine(n_chains=10, sigma=[0.1] * 65, weight=[1 / 65] * 65)
    e.close()


def test_per_chain_beta_with_mixed_pool_and_estimator(gpu, oracle):
    b = np.random.default_rng(5).uniform(0.5, 4.0, 4097)
    e, o = run_pair(gpu, oracle, 4097, 3, "harmonic", 12, beta_arr=b)
    assert_same(e, o)
    g, go = e.pg_estimate([0, 2], 2), o.pg_estimate([0, 2], 2)
    np.testing.assert_allclose(g, go, rtol=1e-10, atol=1e-9)
    assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))
    e.close()


@pytest.mark.parametrize("M,n,counters,K", [(1_000_003, 1, False, 1), (1_000_003, 3, False, 1), (5001, 1, True, 1), (5001, 2, True, 2)])
def test_fused_sweep_and_callback_reduction(gpu, oracle, M, n, counters, K):
    """amc_sweep_reduce_begin: the callback sums of the state AFTER the sweep, formed inside the sweep launch
    (K = 1 pool-wide-counter mode) or by the ordinary two-pass reduction behind it (other modes)."""
    sigma, weight = POOLS[K]
    kw = dict(potential="double_well", beta=2.0, sigma=sigma, weight=weight, seed=13)
    e = gpu.HipEngine(n_chains=M, per_chain_counters=counters, **kw)
    o = oracle.OracleSim(M, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    e.sweep(2)
    o.make_steps(2, 4)
    e.sweep_reduce_begin(n)
    e.sweep(1)                                  # work queued behind the reduction must not disturb it
    red = e.reduce_end()
    o.make_steps(n, 4)
    assert red[3] == M
    np.testing.assert_allclose(red[0] / M, o.energy(), rtol=RED_RTOL)
    mom = o.moments()
    np.testing.assert_allclose(red[1], mom[0], rtol=1e-9, atol=1e-9 * M)
    np.testing.assert_allclose(red[2], mom[1], rtol=RED_RTOL)
    np.testing.assert_allclose(red[4:] / M, o.acceptance(), rtol=RED_RTOL, equal_nan=True)
    o.make_steps(1, 4)
    assert np.array_equal(bits(e.download_state()[0]), bits(o.state()[0]))
    e.sweep_reduce_begin(1)
    e.sweep_reduce_begin(1)                     # two reductions may be in flight ...
    with pytest.raises(gpu.AmcError, match="already in flight"):
        e.sweep_reduce_begin(1)                 # ... a third may not
    e.reduce_end()
    e.reduce_end()
    e.close()


@pytest.mark.parametrize("do_update", [False, True])
@pytest.mark.parametrize("case", ["k3", "k1", "k1_pooled", "k2_beta", "k5_wide", "custom"])
def test_pgmc_steps_one_call_equals_separate_calls(gpu, do_update, case):
    """amc_pgmc_steps(n) == n x [amc_sweep(1); amc_pg_accumulate; amc_pg_update] bit for bit
    (src/simulation.jl:185-190 runs the three algorithms back to back at every t).  With per-chain counters and at most
    two learnable moves the sweep rides in the estimator launch (one HBM round trip of x per time step): K = 1 and
    K > 1 forms, per-chain beta, a run-time compiled potential; "k5_wide" (three learnable moves) takes two launches."""
    from montecarlo_amd import CustomPotential
    M = 40001
    kw = dict(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1, 0.3], weight=[0.5, 0.25, 0.25], seed=19)
    ids, kinds, h0, h1 = [1, 2], [1, 4], [0.5, 0.01], [0.0, 1e-6]            # VPG(0.5), NPG(0.01, 1e-6)
    beta = None
    if case in ("k1", "k1_pooled"):
        kw.update(sigma=[0.15], weight=[1.0])
        ids, kinds, h0, h1 = [0], [2], [0.3], [0.0]                          # BLPG(0.3)
        if case == "k1_pooled":                                              # pool-wide counter only: no step log to ride on
            kw.update(per_chain_counters=False)
    elif case == "k2_beta":
        kw.update(sigma=[0.2, 0.4], weight=[0.5, 0.5], potential="double_well")
        ids, kinds, h0, h1 = [1], [1], [0.2], [0.0]
        beta = np.random.default_rng(5).uniform(0.5, 3.0, M)
    elif case == "k5_wide":
        kw.update(sigma=[0.2, 0.1, 0.3, 0.25, 0.5], weight=[0.2] * 5)
        ids, kinds, h0, h1 = [0, 2, 4], [1, 1, 6], [0.1, 0.1, 1e-6], [0.0, 0.0, 1e-6]
    elif case == "custom":
        kw.update(potential=CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"), sigma=[0.3, 0.6], weight=[0.5, 0.5])
        ids, kinds, h0, h1 = [0, 1], [1, 1], [0.05, 0.05], [0.0, 0.0]
    a, b = gpu.HipEngine(**kw), gpu.HipEngine(**kw)
    x0 = np.random.default_rng(6).uniform(-2, 2, M)
    for e in (a, b):
        if beta is not None:
            e.upload_state(x0, beta)
        else:
            e.init_uniform(-2, 2)
        e.sweep(2)
    n = 37                                                                   # crosses a step-log fold
    a.pgmc_steps(n, ids, 2, kinds if do_update else None, h0, h1)
    for _ in range(n):
        b.sweep(1)
        b.pg_accumulate(ids, 2)
        if do_update:
            b.pg_update(ids, kinds, h0, h1)
    assert np.array_equal(bits(a.download_state()[0]), bits(b.download_state()[0]))
    assert np.array_equal(bits(a.pg_get_accumulated(ids)), bits(b.pg_get_accumulated(ids)))
    for k in range(len(kw["sigma"])):
        assert a.get_parameters(k)[0] == b.get_parameters(k)[0]
    if do_update:
        assert a.get_parameters(ids[0])[0] != kw["sigma"][ids[0]]
    if case != "k1_pooled":
        acc_a, tot_a = a.download_counters()
        acc_b, tot_b = b.download_counters()
        assert np.array_equal(acc_a, acc_b) and np.array_equal(tot_a, tot_b)
    assert np.array_equal(a.counter_totals()[0], b.counter_totals()[0])
    assert np.array_equal(a.counter_totals()[1], b.counter_totals()[1])
    assert a.step == b.step == 2 + n and a.estimator_step == b.estimator_step == n
    np.testing.assert_allclose(a.reduce(), b.reduce(), rtol=1e-13, equal_nan=True)
    a.close(); b.close()


# ---- the accept filter (amc_kernels.h accept_filter / mh_pair) --------------------------------
def test_accept_filter_estimate_stays_inside_its_error_budget(gpu):
    """EVERY float in [-17, 0] (1.1e9 values): the float estimate v_exp_f32(t * log2e) deviates from the spec's f64
    exp(t) by less than the 1.9e-6 the error budget grants the float product and the hardware exponential together
    (eps = 2^-16 = 1.5e-5 is the interval the kernel actually uses)."""
    worst = gpu.selftest_accept_filter(0.0, -17.0)
    assert 0.0 < worst < 1.9e-6, worst
    assert gpu.selftest_accept_filter(0.0, -1e-3) < 2e-7            # near 0 the estimate is a float rounding of ~1
    assert gpu.selftest_accept_filter(-17.0, -1e30) < 1.9e-6         # below -17 the argument is clamped


@pytest.mark.parametrize("K,potential,counters", [(1, "harmonic", False), (2, "double_well", True)])
def test_filtered_and_exact_accept_decisions_agree_on_millions_of_steps(gpu, monkeypatch, K, potential, counters):
    """The same run with the filter (default) and with every decision taken by the reference-ordered arithmetic
    (AMC_EXACT_ACCEPT=1): states and accept counts bit for bit.  3e6 chains x 804 steps = 2.4e9 decisions; ~1e5 of them
    (K = 1: 12-bit bracket of u) fall inside the filter's interval and take the exact path in the filtered run as well."""
    sigma, weight = POOLS[K]
    kw = dict(n_chains=3_000_001, potential=potential, beta=2.0, sigma=sigma, weight=weight, seed=123, per_chain_counters=counters)
    runs = []
    for exact in ("0", "1"):
        monkeypatch.setenv("AMC_EXACT_ACCEPT", exact)
        e = gpu.HipEngine(**kw)
        e.init_uniform(-2, 2)
        for _ in range(4):
            e.sweep(1)
        e.sweep(800)
        runs.append((e.download_state()[0], e.counter_totals()[0]))
        e.close()
    assert np.array_equal(bits(runs[0][0]), bits(runs[1][0]))
    assert np.array_equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("sigma", [1e-100, 1e-9, 3e-4, 25.0, 1e6, 1e100])
def test_accept_filter_at_extreme_step_sizes(gpu, oracle, sigma):
    """dlogp ~ 0 (tiny steps: every decision is inside the |dlogp| <= 1e-12 band or next to it), dlogp << -17 and
    overflowing potentials (huge steps): filter and oracle must still agree bit for bit."""
    M = 50001
    kw = dict(potential="double_well", beta=2.0, sigma=[sigma], weight=[1.0], seed=31)
    e = gpu.HipEngine(n_chains=M, **kw)
    o = oracle.OracleSim(M, **kw)
    x0 = np.random.default_rng(9).uniform(-2, 2, M)
    x0[:7] = [0.0, -0.0, 1.0, -1.0, 1e-200, 1e150, -1e155]        # flat spots of the well, underflow / overflow of x^2
    e.upload_state(x0); o.set_x(x0)
    for _ in range(3):
        e.sweep(1)
    e.sweep(9)
    o.make_steps(12, 8)
    assert_same(e, o)
    e.close()


def test_randomised_configurations_against_the_oracle(gpu, oracle):
    """Forty seeded random configurations -- pool size, step sizes over 12 decades, weights, beta (scalar or per chain,
    over 6 decades), potential, chain count and offset, sweepstep, launch pattern -- each bit for bit against the
    oracle: states, energies, per-chain counters, reductions."""
    rng = np.random.default_rng(2026)
    for case in range(40):
        K = int(rng.choice([1, 1, 2, 3, 5]))
        sigma = list(np.exp(rng.uniform(np.log(1e-6), np.log(1e6), K)))
        w = rng.uniform(0.05, 1.0, K)
        weight = list(w / w.sum())
        weight[-1] = 1.0 - sum(weight[:-1])
        M = int(rng.integers(1, 6000))
        offset = 2 * int(rng.integers(0, 2 ** 33))
        beta = float(np.exp(rng.uniform(np.log(1e-3), np.log(1e3))))
        potential = str(rng.choice(["harmonic", "double_well"]))
        sweepstep = int(rng.choice([1, 1, 2, 5]))
        counters = bool(rng.integers(0, 2)) or K > 1
        kw = dict(potential=potential, beta=beta, sigma=sigma, weight=weight, seed=int(rng.integers(0, 2 ** 63)), sweepstep=sweepstep)
        e = gpu.HipEngine(n_chains=M, chain_offset=offset, n_chains_global=offset + M, per_chain_counters=counters, **kw)
        o = oracle.OracleSim(M, chain_offset=offset, **kw)
        x0 = rng.normal(0, 1.5, M) * float(np.exp(rng.uniform(-3, 3)))
        if rng.random() < 0.4:
            b = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), M))
            e.upload_state(x0, b); o.set_x(x0); o.set_beta(b)
        else:
            e.upload_state(x0); o.set_x(x0)
        n1, n2 = int(rng.integers(0, 4)), int(rng.integers(1, 40))
        for _ in range(n1):
            e.sweep(1)
        e.sweep(n2)
        o.make_steps(n1 + n2, 4)
        try:
            assert_same(e, o, counters=counters)
        except AssertionError as err:
            raise AssertionError(f"case {case}: K={K} sigma={sigma} beta={beta} {potential} M={M} sweepstep={sweepstep} "
                                 f"counters={counters} launches={n1}+{n2}: {err}") from None
        finally:
            e.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("K,counters", [(1, False), (2, True)])
def test_split_engine_equals_single_engine(gpu, K, counters, dtype):
    """SplitEngine: the shard as sub-shards on separate streams (their launches overlap).  Per-chain results are those
    of one engine bit for bit (global chain ids); reductions and gradient sums agree to rounding."""
    sigma, weight = POOLS[K]
    kw = dict(n_chains=100_003, potential="double_well", beta=2.0, sigma=sigma, weight=weight, seed=8, per_chain_counters=counters,
              dtype=dtype)
    one = gpu.HipEngine(**kw)
    parts = gpu.SplitEngine(n_parts=3, **kw)
    assert [p.n_chains for p in parts.parts] == [33334, 33334, 33335] and sum(p.n_chains for p in parts.parts) == 100_003
    for e in (one, parts):
        e.init_uniform(-2, 2)
        for _ in range(3):
            e.sweep(1)
        e.sweep(7)
    assert np.array_equal(bits(one.download_state()[0]), bits(parts.download_state()[0]))
    assert np.array_equal(one.counter_totals()[0], parts.counter_totals()[0])
    if counters:
        assert np.array_equal(one.download_counters()[0], parts.download_counters()[0])
    np.testing.assert_allclose(one.reduce(), parts.reduce(), rtol=1e-12)
    for e in (one, parts):
        e.sweep_reduce_begin(1)
    np.testing.assert_allclose(one.reduce_end(), parts.reduce_end(), rtol=1e-12)
    np.testing.assert_allclose(one.pg_estimate([0], 2), parts.pg_estimate([0], 2), rtol=1e-11)
    assert np.array_equal(bits(one.download_state()[0]), bits(parts.download_state()[0]))
    one.close(); parts.close()


def test_metropolis_streams_option(gpu, tmp_path):
    """Metropolis(streams=2) through the host mirror: same callback rows and final state as streams=1."""
    import montecarlo_amd as ma
    out = []
    for streams in (1, 2):
        chains = ma.ParticleChains.uniform(50_001, 2.0, -2.0, 2.0)
        pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 1.0),)
        al = (dict(algorithm=ma.Metropolis, pool=pool, seed=3, streams=streams),
              dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=ma.build_schedule(60, 10, 10)))
        sim = ma.Simulation(chains, al, 60, path=str(tmp_path / str(streams)))
        ma.run(sim)
        out.append((chains.x.copy(), [v for _, v in sim.algorithms[1].rows[0]], pool[0].accepted_calls))
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and out[0][2] == out[1][2]
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-12)


def test_split_engine_snapshots_and_pool_totals(gpu):
    kw = dict(n_chains=10_001, potential="harmonic", beta=2.0, sigma=[0.2], weight=[1.0], seed=4, per_chain_counters=False)
    one, parts = gpu.HipEngine(**kw), gpu.SplitEngine(n_parts=2, **kw)
    for e in (one, parts):
        e.init_uniform(-2, 2)
        e.sweep(5)
    assert np.array_equal(bits(one.download_strided(3, 7, 1400)), bits(parts.download_strided(3, 7, 1400)))
    assert np.array_equal(one.histogram(-2, 2, 50), parts.histogram(-2, 2, 50))
    parts.set_counter_totals(1234, 5)
    assert parts.counter_totals()[0][0] == 1234 and parts.counter_totals()[1][0] == 5 * 10_001
    one.close(); parts.close()


# ---- round 3: callback sums inside the fused PGMC launch, K - 1 total arrays, 32-bit counter guard -------------------
@pytest.mark.parametrize("case", ["k2", "k1", "k1_pooled", "k2_beta", "k5_wide", "custom", "f32", "scaled", "script"])
def test_pgmc_steps_reduce_begin_equals_steps_then_reduce(gpu, case):
    """amc_pgmc_steps_reduce_begin(n) == amc_pgmc_steps(n); amc_reduce_begin: the callback observes the state AFTER the
    estimator's samples (run! order, src/simulation.jl:185-190).  Fused forms (<= 2 learnable moves) form the sums over
    x inside the last launch and the ratios in the fold of the step log behind it; "k5_wide" takes the plain passes.  x, GradientData, sigma, counters bit for bit; the sums to reduction tolerance; work queued behind the
    reduction does not disturb it."""
    from montecarlo_amd import CustomPotential
    M = 40_003
    kw = dict(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=23)
    ids, kinds, h0, h1 = [1], [1], [0.3], [0.0]
    beta = None
    if case in ("k1", "k1_pooled"):
        kw.update(sigma=[0.15], weight=[1.0], per_chain_counters=(case == "k1"))
        ids = [0]
    elif case == "k2_beta":
        kw.update(potential="double_well")
        ids, kinds, h0, h1 = [0, 1], [1, 2], [0.2, 0.2], [0.0, 0.0]
        beta = np.random.default_rng(5).uniform(0.5, 3.0, M)
    elif case == "k5_wide":
        kw.update(sigma=[0.2, 0.1, 0.3, 0.25, 0.5], weight=[0.2] * 5)
        ids, kinds, h0, h1 = [0, 2, 4], [1, 1, 6], [0.1, 0.1, 1e-6], [0.0, 0.0, 1e-6]
    elif case == "custom":
        kw.update(potential=CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"))
    elif case == "f32":                       # Particle{Float32}: the same sources compiled at run time for the other state type
        kw.update(dtype="f32")
    elif case == "scaled":                    # state-dependent width sigma * scale(x): every decision in the reference's arithmetic
        kw.update(scale_expr="0.5 + x*x")
    elif case == "script":                    # a script-defined proposal (Langevin step) with its sigma-derivative
        kw.update(proposal=("-2.0*sigma*sigma*x + sigma*z",
                            "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)",
                            "((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(sigma*sigma*sigma) "
                            "- 4.0*x*(delta + 2.0*sigma*sigma*x)/sigma - 1.0/sigma"))
    a, b = gpu.HipEngine(**kw), gpu.HipEngine(**kw)
    x0 = np.random.default_rng(6).uniform(-2, 2, M)
    if case == "f32":
        x0 = x0.astype(np.float32).astype(np.float64)
    for e in (a, b):
        e.upload_state(x0, beta)
        e.sweep(3)
    for n in (1, 7):
        a.pgmc_steps(n, ids, 2, kinds, h0, h1, reduce_begin=True)
        a.pgmc_steps(2, ids, 2, kinds, h0, h1)            # queued behind the reduction
        red_a = a.reduce_end()
        b.pgmc_steps(n, ids, 2, kinds, h0, h1)
        red_b = b.reduce()
        b.pgmc_steps(2, ids, 2, kinds, h0, h1)
        assert red_a[3] == red_b[3] == M
        np.testing.assert_allclose(red_a, red_b, rtol=1e-12, equal_nan=True)
    assert np.array_equal(bits(a.download_state()[0]), bits(b.download_state()[0]))
    assert np.array_equal(bits(a.pg_get_accumulated(ids)), bits(b.pg_get_accumulated(ids)))
    assert [a.get_parameters(k)[0] for k in range(len(kw["sigma"]))] == [b.get_parameters(k)[0] for k in range(len(kw["sigma"]))]
    assert np.array_equal(a.counter_totals()[0], b.counter_totals()[0]) and np.array_equal(a.counter_totals()[1], b.counter_totals()[1])
    if case != "k1_pooled":
        ca, cb = a.download_counters(), b.download_counters()
        assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
        assert np.all(ca[1].sum(axis=0) == 3 + 8 + 4)          # every chain took every step
    a.pgmc_steps(1, ids, 2, kinds, h0, h1, reduce_begin=True)
    a.pgmc_steps(1, ids, 2, kinds, h0, h1, reduce_begin=True)            # two reductions may be in flight
    with pytest.raises(gpu.AmcError, match="already in flight"):
        a.pgmc_steps(1, ids, 2, kinds, h0, h1, reduce_begin=True)
    a.reduce_end()
    a.reduce_end()
    a.close(); b.close()


@pytest.mark.parametrize("lag", [1, 2])
@pytest.mark.parametrize("depth", [2, 5, 16])
@pytest.mark.parametrize("K", [1, 2, 3])
def test_callback_folds_interleaved_with_queued_sweeps(gpu, oracle, monkeypatch, depth, K, lag):
    """A callback's fold of the step log (amc_sweep_reduce_begin) with sweeps queued behind it before its sums are read:
    callbacks read `lag` periods late -- one (the pipelined form) or two (TWO reductions pending in the engine, handed back
    oldest first: what StoreCallbacks does with a callback at every step) -- and at once, small log depths so that full
    logs are folded in between; counters and sums against the oracle throughout."""
    monkeypatch.setenv("AMC_LOG_DEPTH", str(depth))
    M = 9001
    sigma, weight = POOLS[K]
    kw = dict(potential="double_well", beta=2.0, sigma=sigma, weight=weight, seed=31)
    e = gpu.HipEngine(n_chains=M, per_chain_counters=True, **kw)
    o = oracle.OracleSim(M, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    expect, got = [], []
    pending = 0
    for i, n in enumerate([1, 3, 1, 4, 2, 7, 1, 1, 5, 3, 2, 6]):
        e.sweep(n)
        o.make_steps(n, 4)
        if pending == lag:
            got.append(e.reduce_end())                    # the oldest pending callback's sums, `lag` periods late
            pending -= 1
        e.sweep_reduce_begin(1 + i % 2)
        o.make_steps(1 + i % 2, 4)
        expect.append((o.energy(), o.moments(), o.acceptance()))
        pending += 1
        if i % 4 == 3:
            while pending:
                got.append(e.reduce_end())                # ... or at once
                pending -= 1
            if i == 7:
                a2, t2 = e.download_counters()            # a reader of the counters between two callbacks
                ao, to = o.counters()
                assert np.array_equal(a2, ao) and np.array_equal(t2, to)
    while pending:
        got.append(e.reduce_end())
        pending -= 1
    assert len(got) == len(expect)
    for red, (en, mom, acc) in zip(got, expect):
        assert red[3] == M
        np.testing.assert_allclose(red[0] / M, en, rtol=RED_RTOL)
        np.testing.assert_allclose(red[2], mom[1], rtol=RED_RTOL)
        np.testing.assert_allclose(red[4:] / M, acc, rtol=RED_RTOL, equal_nan=True)
    assert_same(e, o)
    e.close()


@pytest.mark.parametrize("K", [1, 2, 3])
def test_per_chain_counters_count_past_2_to_the_32(gpu, oracle, K):
    """Move.accepted_calls / total_calls are Int64 in the reference (src/metropolis.jl:145-146).  The device counts in 32-bit
    arrays and carries them into 64-bit bases before the launch that would count step 2^32 (round 5; until round 4 that call was
    refused): a run whose count starts five steps short of 2^32 keeps counting -- counters, pool totals, callback_acceptance's
    ratio sums (formed from arrays plus bases after the carry) and a fused PGMC stretch equal the oracle's, whose counters are
    64-bit all along; a checkpoint taken beyond 2^32 goes back in."""
    M = 1001
    sigma, weight = {1: ([0.1], [1.0]), 2: ([0.2, 0.1], [0.6, 0.4]), 3: ([0.3, 0.2, 0.1], [0.5, 0.3, 0.2])}[K]
    kw = dict(n_chains=M, potential="harmonic", beta=2.0, sigma=sigma, weight=weight, seed=5, per_chain_counters=True)
    e, o = gpu.HipEngine(device=0, **kw), oracle.OracleEngine(**kw)
    rng = np.random.default_rng(K)
    start = 2 ** 32 - 6
    tot = np.zeros((K, M), dtype=np.int64)
    cuts = np.sort(rng.integers(0, start + 1, size=(K - 1, M)), axis=0) if K > 1 else np.zeros((0, M), dtype=np.int64)
    edges = np.vstack([np.zeros((1, M), dtype=np.int64), cuts, np.full((1, M), start, dtype=np.int64)])
    tot[:] = np.diff(edges, axis=0)                         # every chain's totals add up to `start`
    acc = (tot * rng.uniform(0.3, 1.0, size=tot.shape)).astype(np.int64)
    for eng in (e, o):
        eng.init_uniform(-2, 2)
        eng.upload_counters(acc, tot)

    def same(what):
        a, t = e.download_counters()
        ao, to = o.download_counters()
        assert np.array_equal(a, ao) and np.array_equal(t, to), what
        assert np.array_equal(e.counter_totals()[0], ao.sum(axis=1)) and np.array_equal(e.counter_totals()[1], to.sum(axis=1)), what
        ra, sa = e.reduce_exact()
        rb, sb = o.reduce_exact()
        assert sa == sb == int(t[:, 0].sum()) and np.array_equal(ra, rb), what
    same("before the carry")
    for n in (3, 4, 2, 70):                                 # 3: still below; 4: the launch that crosses 2^32 carries first
        e.sweep(n)
        o.sweep(n)
        same(f"after {n} more")
    assert e.download_counters()[1].sum(axis=0).min() == start + 79 > 2 ** 32
    # the callback sums formed with a sweep, and a fused PGMC stretch, on a handle that has carried
    e.sweep_reduce_begin(3)
    o.sweep_reduce_begin(3)
    ra, sa = e.reduce_end_exact()
    rb, sb = o.reduce_end_exact()
    assert sa == sb and np.array_equal(ra, rb)
    lid = K - 1
    e.pgmc_steps(5, [lid], 1, [1], [1e-3], [0.0], reduce_begin=True)
    o.pgmc_steps(5, [lid], 1, [1], [1e-3], [0.0], reduce_begin=True)
    ra, sa = e.reduce_end_exact()
    rb, sb = o.reduce_end_exact()
    assert sa == sb and np.array_equal(ra, rb) and np.array_equal(bits(e.get_parameters(lid)), bits(o.get_parameters(lid)))
    same("after the fused stretch")
    # resume: counters beyond 2^32 go back in, on a fresh handle too
    a, t = e.download_counters()
    x = e.download_state()[0]
    e2 = gpu.HipEngine(device=0, **kw)
    e2.upload_state(x)
    e2.upload_counters(a, t)
    e2.step = e.step
    e2.set_parameters(lid, e.get_parameters(lid))
    for eng in (e, e2, o):
        eng.sweep(11)
    a2, t2 = e2.download_counters()
    ao, to = o.download_counters()
    assert np.array_equal(a2, ao) and np.array_equal(t2, to)
    same("after the resume")
    # ... and counts that fit 32 bits again restart plain arrays
    e2.upload_counters(acc * 0, tot * 0)
    e2.sweep(4)
    assert np.all(e2.download_counters()[1].sum(axis=0) == 4)
    with pytest.raises(gpu.AmcError, match="out of range"):
        e2.upload_counters(acc * 0 + 2 ** 53, None if K == 1 else tot)
    # the pool-wide counter of a K = 1 handle without per-chain counters is 64-bit
    if K == 1:
        p = gpu.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=sigma, weight=weight, seed=5, per_chain_counters=False)
        p.init_uniform(-2, 2)
        p.set_counter_totals(7 * 2 ** 40, 2 ** 41)
        p.sweep(5)
        a, t = p.counter_totals()
        assert t[0] == (2 ** 41 + 5) * M and 7 * 2 ** 40 <= a[0] <= 7 * 2 ** 40 + 5 * M
        p.close()
    e.close()
    e2.close()


@pytest.mark.parametrize("K", [1, 2, 3])
def test_counters_across_the_16_bit_mark(gpu, oracle, K):
    """Handles with K <= 4 keep their per-chain counters as two u16 planes; the high plane stays out of the folds until the call
    that would count past 65 535 steps (amc_api.hip counter_room) and is written only where a low half carries: counts,
    acceptance sums and the states on both sides of the mark against the oracle's increments, with neighbouring counters at
    0xFFFF and 0 (the packing of the 16-bit quads, carries in some lanes of a quad only)."""
    M = 2051
    sigma, weight = POOLS[K]
    kw = dict(potential="double_well", beta=2.0, sigma=sigma, weight=weight, seed=11)
    e = gpu.HipEngine(n_chains=M, per_chain_counters=True, **kw)
    o = oracle.OracleSim(M, **kw)
    e.init_uniform(-2, 2)
    o.init_uniform(-2, 2)
    base = 65530
    rng = np.random.default_rng(2)
    tot = rng.multinomial(base, weight, size=M).T.astype(np.int64)
    tot[:, 0::4] = 0
    tot[0, 0::4] = base                                     # every fourth chain: one counter at the top, the others at 0
    acc = (tot * rng.uniform(0, 1, tot.shape)).astype(np.int64)
    acc[0, 0::8] = base
    e.upload_counters(acc, tot)
    a1, t1 = e.download_counters()
    assert np.array_equal(a1, acc) and np.array_equal(t1, tot)

    def check(steps_done):
        ao, to = o.counters()
        a, t = e.download_counters()
        assert np.array_equal(a, acc + ao) and np.array_equal(t, tot + to), steps_done
        assert np.array_equal(e.counter_totals()[0], (acc + ao).sum(axis=1))
        assert np.array_equal(e.counter_totals()[1], (tot + to).sum(axis=1))
        red = e.reduce()
        with np.errstate(invalid="ignore", divide="ignore"):
            np.testing.assert_allclose(red[4:], ((acc + ao) / (tot + to)).sum(axis=1), rtol=1e-12, equal_nan=True)
        assert np.array_equal(e.download_state()[0].view(np.uint64), o.state()[0].view(np.uint64))

    e.sweep(3); o.make_steps(3)                             # 65 533: still 16-bit
    check(3)
    e.sweep(2); o.make_steps(2)                             # 65 535: the last value u16 holds
    check(5)
    e.sweep(1); o.make_steps(1)                             # widened before this step
    check(6)
    e.sweep_reduce_begin(40); e.reduce_end(); o.make_steps(40)
    check(46)
    # a fused PGMC call that crosses the mark widens up front as well
    e2 = gpu.HipEngine(n_chains=M, per_chain_counters=True, **kw)
    o2 = oracle.OracleSim(M, **kw)
    e2.init_uniform(-2, 2); o2.init_uniform(-2, 2)
    e2.upload_counters(acc, tot)
    e2.sweep(4); o2.make_steps(4)                           # rows waiting in the log when the widening happens
    e2.pgmc_steps(3, [0], 1)
    for _ in range(3):
        o2.make_steps(1)
        o2.pg_estimate([0], 1)
    ao, to = o2.counters()
    a, t = e2.download_counters()
    assert np.array_equal(a, acc + ao) and np.array_equal(t, tot + to)
    assert np.array_equal(e2.download_state()[0].view(np.uint64), o2.state()[0].view(np.uint64))
    # counts that do not fit 16 bits go straight to the wide arrays, and back when the count restarts
    e2.upload_counters(acc * 3, tot * 3)
    assert np.array_equal(e2.download_counters()[1], tot * 3)
    e2.upload_counters(acc * 0, tot * 0)
    e2.sweep(7)
    assert np.all(e2.download_counters()[1].sum(axis=0) == 7)
    e.close(); e2.close()


def test_histogram_accumulated_on_the_device(gpu, oracle):
    """amc_histogram_accumulate / _fetch: the running histogram over several sample times equals the sum of the one-shot
    histograms (and the oracle's), stays on the device between calls, and is re-binned only after a reset."""
    M = 20_011
    kw = dict(potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=4)
    e = gpu.HipEngine(n_chains=M, **kw)
    o = oracle.OracleEngine(n_chains=M, **kw)
    e.init_uniform(-2.5, 2.5); o.init_uniform(-2.5, 2.5)
    want = np.zeros(43, dtype=np.uint64)
    for _ in range(4):
        e.sweep(3); o.sweep(3)
        e.histogram_accumulate(-2.0, 2.0, 40)
        want += o.histogram(-2.0, 2.0, 40)
    with pytest.raises(gpu.AmcError, match="other bins"):
        e.histogram_accumulate(-2.0, 2.0, 41)
    got = e.histogram_fetch(40, reset=False)
    assert np.array_equal(got, want) and got.sum() == 4 * M
    e.sweep(2); o.sweep(2)
    e.histogram_accumulate(-2.0, 2.0, 40)
    want += o.histogram(-2.0, 2.0, 40)
    assert np.array_equal(e.histogram_fetch(40), want)                 # reset
    with pytest.raises(gpu.AmcError, match="nothing has been accumulated"):
        e.histogram_fetch(40)
    e.histogram_accumulate(-1.0, 1.0, 8)                               # other bins after the reset
    assert np.array_equal(e.histogram_fetch(8), o.histogram(-1.0, 1.0, 8)) and np.array_equal(e.histogram(-1.0, 1.0, 8), o.histogram(-1.0, 1.0, 8))
    e.close(); o.close()


def test_parameters_read_in_stream_order_without_draining_the_queue(gpu, oracle):
    """amc_parameters_begin / _end (StoreParameters beside device-resident learning steps, metropolis.jl:433-440): the read is a
    copy queued at its point of the stream -- steps queued AFTER it do not show in it -- fetched later; one in flight."""
    M = 4099
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=3)
    e = gpu.HipEngine(n_chains=M, **kw)
    o = oracle.OracleEngine(n_chains=M, **kw)
    e.init_uniform(-2, 2); o.init_uniform(-2, 2)
    vpg = [oracle.OPTIMISERS["VPG"]]
    e.pgmc_steps(5, [1], 2, vpg, [0.05], [0.0]); o.pgmc_steps(5, [1], 2, vpg, [0.05], [0.0])
    e.parameters_begin()
    with pytest.raises(gpu.AmcError, match="already in flight"):
        e.parameters_begin()
    e.pgmc_steps(7, [1], 2, vpg, [0.05], [0.0])                        # queued behind the read
    s5 = e.parameters_end()
    assert s5[0] == 0.2 and abs(s5[1] - o.get_parameters(1)[0]) < 1e-10 * s5[1] and s5[1] != 0.1
    with pytest.raises(gpu.AmcError, match="no read in flight"):
        e.parameters_end()
    o.pgmc_steps(7, [1], 2, vpg, [0.05], [0.0])
    e.parameters_begin()
    s12 = e.parameters_end()
    assert abs(s12[1] - o.get_parameters(1)[0]) < 1e-9 * s12[1] and s12[1] != s5[1] and s12[1] == e.get_parameters(1)[0]
    e.close(); o.close()


def test_uploaded_totals_must_add_up_to_one_step_count(gpu):
    """Every chain takes the same number of MH steps (mc_sweep!, metropolis.jl:205-210): sum_k total_calls is one number
    for all chains, and the device keeps K - 1 of the K total arrays."""
    M, K = 501, 3
    sigma, weight = POOLS[K]
    e = gpu.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=sigma, weight=weight, seed=5)
    e.init_uniform(-2, 2)
    rng = np.random.default_rng(1)
    tot = rng.multinomial(1000, weight, size=M).T.astype(np.int64)
    acc = (tot * rng.uniform(0, 1, tot.shape)).astype(np.int64)
    e.upload_counters(acc, tot)
    a2, t2 = e.download_counters()
    assert np.array_equal(a2, acc) and np.array_equal(t2, tot)
    assert np.array_equal(e.counter_totals()[1], tot.sum(axis=1))
    red = e.reduce()
    with np.errstate(invalid="ignore", divide="ignore"):
        np.testing.assert_allclose(red[4:], (acc / tot).sum(axis=1), rtol=1e-12, equal_nan=True)
    e.sweep(5)
    assert np.all(e.download_counters()[1].sum(axis=0) == 1005)
    bad = tot.copy()
    bad[1, 7] += 1
    with pytest.raises(gpu.AmcError, match="same step count"):
        e.upload_counters(acc, bad)
    e.close()


def test_estimator_entry_points_refuse_bad_arguments(gpu):
    """The C ABI's own checks (include/amc.h: negative status + amc_last_error, nothing throws or faults): a NULL id array, counts out
    of range, ids that name no move -- on the handle kinds whose routes read the ids on the host before a launch would look at them."""
    import ctypes as C
    lib = gpu.load()
    drift = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)", None)
    for kw in (dict(sigma=[0.2, 0.4], weight=[0.5, 0.5]),
               dict(sigma=[[0.0, 0.5], [0.1, 0.9]], weight=[0.5, 0.5], proposal=drift, n_params=2),
               dict(sigma=[0.1] * 7, weight=[0.4] + [0.1] * 6)):
        e = gpu.HipEngine(n_chains=1001, potential="harmonic", beta=2.0, seed=3, **kw)
        e.init_uniform(-2, 2)
        h = e._h
        ids = (C.c_int * 8)(0, 1, 0, 0, 0, 0, 0, 0)
        bad = (C.c_int * 2)(0, 99)
        assert lib.amc_pg_accumulate(h, 2, None, 1) == -1 and b"learn_ids is NULL" in lib.amc_last_error()
        assert lib.amc_pg_accumulate(h, 1000, ids, 1) == -1 and lib.amc_pg_accumulate(h, -1, ids, 1) == -1
        assert lib.amc_pg_accumulate(h, 2, bad, 1) == -1 and b"out of range" in lib.amc_last_error()
        assert lib.amc_pg_accumulate(h, 2, ids, 0) == -1
        assert lib.amc_pgmc_steps(h, 1, 2, None, 1, 0, None, None, None) == -1
        assert lib.amc_pgmc_steps(h, 1, 2, ids, 1, 1, None, None, None) == -1            # an update without optimisers
        x_before, t_before = e.download_state()[0], e.estimator_step
        out = (C.c_double * 64)()
        assert lib.amc_pg_estimate(h, 2, bad, 1, out) == -1 and b"out of range" in lib.amc_last_error()
        assert lib.amc_pg_estimate(h, 2, None, 1, out) == -1
        assert np.array_equal(e.download_state()[0], x_before) and e.estimator_step == t_before      # nothing was drawn
        assert lib.amc_pg_accumulate(h, 0, None, 1) == 0                                 # no learnable move: the estimator step counts on
        e.pg_accumulate([0, 1], 2)                                                       # and the handle is as usable as before
        assert e.pg_get_accumulated([0, 1])[0, -1] == 2 * 1001
        e.close()


def test_strided_snapshot_ranges(gpu):
    """amc_download_strided: every range that leaves the shard is refused -- the ones whose last index would overflow an Int64
    included -- and the edges of the shard are served."""
    import ctypes as C
    lib = gpu.load()
    M = 1001
    e = gpu.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.3], weight=[1.0], seed=5)
    e.init_uniform(-2, 2)
    x = e.download_state()[0]
    assert np.array_equal(e.download_strided(0, 1, M), x) and np.array_equal(e.download_strided(M - 1, 1, 1), x[-1:])
    assert np.array_equal(e.download_strided(0, 500, 3), x[0::500]) and e.download_strided(5, 1, 0).size == 0
    out = (C.c_double * 8)()
    for first, stride, count in ((0, 1, M + 1), (M, 1, 1), (-1, 1, 1), (0, 0, 1), (0, 1, -1), (1, 500, 3), (0, 2**62, 3), (0, 4, 2**62), (2**62, 1, 1),
                                 (0, 2**63 - 1, 2)):
        assert lib.amc_download_strided(e._h, first, stride, count, out) == -1, (first, stride, count)
    e.close()
