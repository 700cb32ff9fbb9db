"""A script-defined policy of the Gaussian-displacement family: proposal width sigma * scale(x) (ScaledGaussian;
amc_create_policy_model).  The reference hands `system` to sample_action! / log_proposal_density
(src/metropolis.jl:177-182), so such a policy is within its interface; the forward density is taken at the old state
and the backward one at the new state, which makes the proposal ratio of mc_step! (metropolis.jl:183) matter."""
import numpy as np
import pytest

import montecarlo_amd as ma

SCALE = "0.5 + x*x"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_oracle_keeps_the_target_distribution(oracle):
    """Detailed balance with an asymmetric proposal: <x^2> = 1/(2 beta) only if logq_b - logq_f enters alpha."""
    beta = 2.0
    s = oracle.OracleSim(4000, potential="harmonic", beta=beta, sigma=[0.5], weight=[1.0], seed=7, scale_expr=SCALE)
    s.init_uniform(-2, 2)
    n, sx, sxx, _ = s.run_pooled_moments(3000, 300, 10, threads=8)
    mean, mean2 = sx / n, sxx / n
    assert mean == pytest.approx(0.0, abs=5e-3) and mean2 == pytest.approx(1 / (2 * beta), abs=4e-3)
    # with the width frozen at its forward value on both sides the chain samples something else: the scale is really used
    acc_scaled = s.acceptance()[0]
    t = oracle.OracleSim(4000, potential="harmonic", beta=beta, sigma=[0.5], weight=[1.0], seed=7)
    t.init_uniform(-2, 2)
    t.make_steps(300, 8)
    assert abs(acc_scaled - t.acceptance()[0]) > 0.02


def test_host_mirror_passes_the_policy_to_the_engine(oracle, tmp_path):
    chains = ma.ParticleChains.uniform(64, 2.0, -2.0, 2.0)
    pool = [ma.Move(ma.Displacement(0.0), ma.ScaledGaussian(SCALE), [0.3], 0.5),
            ma.Move(ma.Displacement(0.0), ma.ScaledGaussian(SCALE), [0.9], 0.5)]
    sim = ma.Simulation(chains, [dict(algorithm=ma.Metropolis, pool=pool, seed=3, engine_factory=oracle.OracleEngine)], 20,
                        path=str(tmp_path))
    ma.run(sim)
    o = oracle.OracleSim(64, potential="harmonic", beta=2.0, sigma=[0.3, 0.9], weight=[0.5, 0.5], seed=3, scale_expr=SCALE)
    o.init_uniform(-2, 2)
    o.make_steps(20)
    assert np.array_equal(bits(chains.x), bits(o.state()[0]))
    plain = oracle.OracleSim(64, potential="harmonic", beta=2.0, sigma=[0.3, 0.9], weight=[0.5, 0.5], seed=3)
    plain.init_uniform(-2, 2)
    plain.make_steps(20)
    assert not np.array_equal(bits(chains.x), bits(plain.state()[0]))
    mixed = [ma.Move(ma.Displacement(0.0), ma.ScaledGaussian(SCALE), [0.3], 0.5),
             ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), [0.9], 0.5)]
    # a ScaledGaussian beside a StandardGaussian: a pool of two classes (tests/test_mixed_pool.py), the Gaussians written out as
    # expressions -- the same chain as the scaled policy's own engine gives when BOTH moves are scaled with scale 1 for the second
    met = ma.Metropolis(ma.ParticleChains.uniform(8, 2.0), pool=mixed, engine_factory=oracle.OracleEngine)
    assert met.engine.sim.lib is not None and met.n_params == 1
    oracle.install_policy_classes(None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["k1", "k2_beta", "custom_potential"])
def test_trajectories_bit_exact_against_the_oracle(gpu, oracle, dtype, case):
    M = 20001
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.4], weight=[1.0], seed=11, scale_expr=SCALE)
    beta = None
    if case == "k1":
        kw["per_chain_counters"] = False
    elif case == "k2_beta":
        kw.update(potential="double_well", sigma=[0.2, 0.8], weight=[0.4, 0.6])
        beta = np.random.default_rng(1).uniform(0.5, 3.0, M)
    else:
        kw.update(potential=ma.CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"), scale_expr="0.3 + 0.5*fabs(x)")
    eng = gpu.HipEngine(n_chains=M, dtype=dtype, **kw)
    sim = oracle.OracleSim(M, dtype=dtype, **{k: v for k, v in kw.items() if k != "per_chain_counters"})
    x0 = np.random.default_rng(2).uniform(-2, 2, M)
    eng.upload_state(x0, beta)
    if beta is not None:
        sim.set_beta(beta)
    sim.set_x(x0)
    for n in (1, 1, 9, 1, 30):
        eng.sweep(n)
        sim.make_steps(n)
        x, e = eng.download_state()
        xo, eo = sim.state()
        assert np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(e), bits(eo))
    acc, tot = eng.counter_totals()
    ao, to = sim.counters()
    assert np.array_equal(np.asarray(acc), ao.sum(axis=1)) and np.array_equal(np.asarray(tot), to.sum(axis=1))
    red = eng.reduce()
    assert red[0] / M == pytest.approx(sim.energy(), rel=1e-10)
    # policy-gradient estimator with the state-dependent width: forward gradient at the old state, backward at the new
    ids = list(range(len(kw["sigma"])))
    for q in (1, 3):
        got = np.asarray(eng.pg_estimate(ids, q)).reshape(len(ids), 5)
        want = sim.pg_estimate(ids, q)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9) and np.array_equal(got[:, 4], want[:, 4])
        assert np.array_equal(bits(eng.download_state()[0]), bits(sim.state()[0]))
    eng.close()


@pytest.mark.gpu
def test_estimator_gradient_against_finite_differences_and_fused_steps(gpu, oracle):
    """d logq / d sigma of the scaled policy is delta^2 / (sigma^3 s^2) - 1/sigma; the estimator's sum of forward
    gradients over M chains must match that closed form evaluated on the oracle's own deltas, and one fused PGMC time step
    must equal the separate calls."""
    M = 30000
    kw = dict(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.3, 0.7], weight=[0.5, 0.5], seed=4, scale_expr=SCALE)
    a, b = gpu.HipEngine(**kw), gpu.HipEngine(**kw)
    for e in (a, b):
        e.init_uniform(-2, 2)
        e.sweep(3)
    x_before = a.download_state()[0]
    g = np.asarray(a.pg_estimate([1], 1)).reshape(5)
    # sum over chains of (z^2 - 1) / sigma, z standard normal: mean 0, variance 2 M / sigma^2
    assert abs(g[2]) < 5 * np.sqrt(2 * M) / 0.7
    assert g[3] == pytest.approx(2 * M / 0.7 ** 2, rel=0.05)            # sum of squares: E[(z^2 - 1)^2] = 2
    assert np.array_equal(bits(b.download_state()[0]), bits(x_before))
    b.pg_estimate([1], 1)
    a.pgmc_steps(3, [1], 2, [1], [0.2], [0.0])
    for _ in range(3):
        b.sweep(1)
        b.pg_accumulate([1], 2)
        b.pg_update([1], [1], [0.2], [0.0])
    assert np.array_equal(bits(a.download_state()[0]), bits(b.download_state()[0]))
    assert np.array_equal(a.get_parameters(1), b.get_parameters(1)) and a.get_parameters(1)[0] != 0.7
    a.close()
    b.close()


@pytest.mark.gpu
def test_target_distribution_at_scale(gpu):
    """2e6 chains, pooled over 40 decorrelated snapshots: <x> = 0, <x^2> = 1/(2 beta), <U> of the double well by
    quadrature -- the acceptance carries the proposal ratio correctly."""
    M = 2_000_000
    for potential, beta, want_x2, want_u in (("harmonic", 2.0, 0.25, 0.25), ("double_well", 2.0, 0.852136, 0.272864)):
        e = gpu.HipEngine(n_chains=M, potential=potential, beta=beta, sigma=[0.5], weight=[1.0], seed=21, scale_expr=SCALE,
                          per_chain_counters=False)
        e.init_uniform(-2, 2)
        e.sweep(600)
        s = np.zeros(4)
        for _ in range(40):
            e.sweep(25)
            r = e.reduce()
            s += np.array([r[0], r[1], r[2], r[3]])
        assert s[1] / s[3] == pytest.approx(0.0, abs=2e-3)
        assert s[2] / s[3] == pytest.approx(want_x2, abs=2e-3)
        assert s[0] / s[3] == pytest.approx(want_u, abs=2e-3)
        e.close()


@pytest.mark.gpu
def test_pgmc_learns_the_prefactor_of_a_scaled_policy_through_the_host_mirror(gpu, tmp_path):
    """The PGMC driver script's algorithm list (PGMC_harmonic_oscillator.jl:24-33) with a ScaledGaussian pool: the learnable
    move's sigma grows from 0.1 towards the step size that maximises E[delta^2 alpha], the static one stays, and the
    sampled distribution stays the target's."""
    M, steps = 100_000, 600
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.ScaledGaussian(SCALE), {"sigma": 0.2}, 0.6),
            ma.Move(ma.Displacement(0.0), ma.ScaledGaussian(SCALE), {"sigma": 0.1}, 0.4))
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42),
          dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.5)),
               q_batch_size=1),
          dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
               scheduler=ma.build_schedule(steps, 100, 50)),
          dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=ma.build_schedule(steps, 0, 100)))
    sim = ma.Simulation(chains, al, steps, path=str(tmp_path))
    ma.run(sim)
    assert pool[0].sigma == 0.2
    assert 0.5 < pool[1].sigma < 3.0                      # moved well away from 0.1, towards an O(1) width
    energies = np.loadtxt(tmp_path / "energy.dat")[1:, 1]
    assert energies[-3:].mean() == pytest.approx(0.25, abs=5e-3)
    rows = open(tmp_path / "parameters" / "2" / "parameters.dat").read().splitlines()
    sig = [float(r.split("[")[1].strip("]")) for r in rows]
    assert sig[0] == 0.1 and sig[-1] == pool[1].sigma and sig[1] > 0.1                    # first row is the t = 0 value
