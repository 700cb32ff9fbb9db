"""A script-defined PROPOSAL in full (ScriptPolicy; amc_create_proposal_model): the model's own sample_action! and
log_proposal_density -- the reference declares them as generic functions the user overloads (src/metropolis.jl:35-62;
example/particle_1d/particle_1d.jl:52-59 are the Gaussian displacement's methods) -- given as C expressions and compiled
for gfx950 at run time, with the oracle evaluating the same expressions compiled by gcc.  The example policy is a
drifted Gaussian (Langevin / MALA) proposal for U = x^2, beta = 2: its mean depends on the state, so log q_backward -
log q_forward is the whole point of mc_step!'s acceptance ratio (metropolis.jl:183)."""
import numpy as np
import pytest

import montecarlo_amd as ma

BETA = 2.0
# delta ~ Normal(-beta sigma^2 x, sigma): (sigma^2 / 2) grad log p(x) + sigma z with log p = -beta x^2
SAMPLE = "-2.0*sigma*sigma*x + sigma*z"
LOGQ = "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)"
DLOGQ = ("((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(sigma*sigma*sigma) "
         "- 4.0*x*(delta + 2.0*sigma*sigma*x)/sigma - 1.0/sigma")
MALA = (SAMPLE, LOGQ, DLOGQ)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_oracle_keeps_the_target_distribution_with_a_drifted_proposal(oracle):
    s = oracle.OracleSim(4000, potential="harmonic", beta=BETA, sigma=[0.6], weight=[1.0], seed=5, proposal=MALA)
    s.init_uniform(-2, 2)
    n, sx, sxx, _ = s.run_pooled_moments(2500, 300, 10, threads=8)
    assert sx / n == pytest.approx(0.0, abs=5e-3) and sxx / n == pytest.approx(1 / (2 * BETA), abs=4e-3)
    acc_mala = s.acceptance()[0]
    # without the drift's density ratio (symmetric random walk of the same width) the acceptance is far lower: the
    # Langevin drift is what keeps sigma = 0.6 above 90 %
    t = oracle.OracleSim(4000, potential="harmonic", beta=BETA, sigma=[0.6], weight=[1.0], seed=5)
    t.init_uniform(-2, 2)
    t.make_steps(300, 8)
    assert acc_mala > t.acceptance()[0] + 0.1
    # a WRONG density (drift left out of logq) visibly breaks the stationary distribution: the ratio is really used
    wrong = oracle.OracleSim(4000, potential="harmonic", beta=BETA, sigma=[0.6], weight=[1.0], seed=5,
                             proposal=(SAMPLE, "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(sigma)", None))
    wrong.init_uniform(-2, 2)
    n, sx, sxx, _ = wrong.run_pooled_moments(2500, 300, 10, threads=8)
    assert abs(sxx / n - 0.25) > 0.02
    oracle.install_custom_proposal(None)


def test_host_mirror_passes_the_script_policy_to_the_engine(oracle, tmp_path):
    chains = ma.ParticleChains.uniform(64, BETA, -2.0, 2.0)
    pol = ma.ScriptPolicy(*MALA)
    pool = [ma.Move(ma.Displacement(0.0), pol, [0.3], 0.5), ma.Move(ma.Displacement(0.0), pol, [0.7], 0.5)]
    sim = ma.Simulation(chains, [dict(algorithm=ma.Metropolis, pool=pool, seed=3, engine_factory=oracle.OracleEngine)], 20,
                        path=str(tmp_path))
    ma.run(sim)
    o = oracle.OracleSim(64, potential="harmonic", beta=BETA, sigma=[0.3, 0.7], weight=[0.5, 0.5], seed=3, proposal=MALA)
    o.init_uniform(-2, 2)
    o.make_steps(20)
    assert np.array_equal(bits(chains.x), bits(o.state()[0]))
    # a pool that mixes policy types is a pool of classes (tests/test_mixed_pool.py)
    mixed = [ma.Move(ma.Displacement(0.0), pol, [0.3], 0.5), ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), [0.9], 0.5)]
    met = ma.Metropolis(ma.ParticleChains.uniform(8, BETA), pool=mixed, engine_factory=oracle.OracleEngine)
    assert met.n_params == 1
    oracle.install_policy_classes(None, None)
    oracle.install_custom_proposal(None)


def test_argument_validation_needs_no_gpu(amc):
    kw = dict(n_chains=10, potential="harmonic", beta=BETA, sigma=[0.5], weight=[1.0])
    with pytest.raises(amc.AmcError, match="both required"):
        amc.HipEngine(proposal=(SAMPLE, None, None), **kw)
    with pytest.raises(amc.AmcError, match="does not mention z"):
        amc.HipEngine(proposal=("sigma*x", LOGQ, None), **kw)
    with pytest.raises(amc.AmcError, match="does not mention delta"):
        amc.HipEngine(proposal=(SAMPLE, "x*x", None), **kw)
    # (Float32 state is no error any more: tests/test_f32_script_policy.py)
    with pytest.raises(amc.AmcError, match="not allowed"):
        amc.HipEngine(proposal=(SAMPLE + "; }", LOGQ, None), **kw)
    with pytest.raises(amc.AmcError, match="cannot be combined"):
        amc.HipEngine(proposal=MALA, scale_expr="1.0 + x*x", **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["k1", "k2_beta", "custom_potential"])
def test_trajectories_bit_exact_against_the_oracle(gpu, oracle, case):
    M = 20001
    kw = dict(potential="harmonic", beta=BETA, sigma=[0.4], weight=[1.0], seed=11, proposal=MALA)
    beta = None
    if case == "k1":
        kw["per_chain_counters"] = False
    elif case == "k2_beta":
        kw.update(potential="double_well", sigma=[0.2, 0.6], weight=[0.4, 0.6])
        beta = np.random.default_rng(1).uniform(0.5, 3.0, M)
    else:
        kw.update(potential=ma.CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"),
                  proposal=("sigma*z*(0.5 + fabs(x))", "-(delta*delta)/(2.0*(sigma*(0.5 + fabs(x)))*(sigma*(0.5 + fabs(x)))) - amc_log(sigma*(0.5 + fabs(x)))",
                            "(delta*delta)/((sigma*sigma*sigma)*(0.5 + fabs(x))*(0.5 + fabs(x))) - 1.0/sigma"))
    eng = gpu.HipEngine(n_chains=M, **kw)
    sim = oracle.OracleSim(M, **{k: v for k, v in kw.items() if k != "per_chain_counters"})
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
    ids = list(range(len(kw["sigma"])))
    for q in (1, 3):
        got = np.asarray(eng.pg_estimate(ids, q)).reshape(len(ids), 5)
        want = sim.pg_estimate(ids, q)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9) and np.array_equal(got[:, 4], want[:, 4])
        assert np.array_equal(bits(eng.download_state()[0]), bits(sim.state()[0]))
    eng.close()
    oracle.install_custom_proposal(None)


@pytest.mark.gpu
def test_estimator_needs_the_sigma_derivative_and_target_distribution_at_scale(gpu):
    M = 2_000_000
    e = gpu.HipEngine(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.6], weight=[1.0], seed=21, proposal=(SAMPLE, LOGQ, None),
                      per_chain_counters=False)
    e.init_uniform(-2, 2)
    # (round 6: a proposal without its derivative runs the estimator all the same -- the engine differentiates logq like the
    # reference's ForwardDiff backend does, gradients.jl:28-33; tests/test_autodiff.py)
    e.sweep(400)
    s = np.zeros(4)
    for _ in range(30):
        e.sweep(20)
        s += e.reduce()[:4]
    assert s[1] / s[3] == pytest.approx(0.0, abs=2e-3) and s[2] / s[3] == pytest.approx(0.25, abs=2e-3)
    red = e.reduce()
    assert red[4] / M > 0.8                                   # a random walk of this width accepts 66 %, the Langevin drift 86 %
    e.close()
    # with the derivative: E[d logq / d sigma] = 0 under the proposal, E[(d logq / d sigma)^2] is the Fisher information 2/sigma^2 + 16 sigma^2 <x^2>
    f = gpu.HipEngine(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.6], weight=[1.0], seed=22, proposal=MALA)
    f.init_uniform(-2, 2)
    f.sweep(300)
    x = f.download_state()[0]
    g = np.asarray(f.pg_estimate([0], 1)).reshape(5)
    # d logq / d sigma = (eps^2 - 1)/sigma - 4 x eps with eps standard normal: variance 2/sigma^2 + 16 <x^2>
    fisher = 2 / 0.6 ** 2 + 16 * float(np.mean(x * x))
    assert abs(g[2]) < 6 * np.sqrt(M * fisher) and g[3] / M == pytest.approx(fisher, rel=0.02)
    # one fused PGMC call (sweep + estimator + learning step per launch) equals the separate calls, bit for bit
    h = gpu.HipEngine(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.6], weight=[1.0], seed=22, proposal=MALA)
    h.upload_state(f.download_state()[0])
    h.step, h.estimator_step = f.step, f.estimator_step
    f.pgmc_steps(3, [0], 2, [1], [0.05], [0.0])
    for _ in range(3):
        h.sweep(1)
        h.pg_accumulate([0], 2)
        h.pg_update([0], [1], [0.05], [0.0])
    assert np.array_equal(bits(f.download_state()[0]), bits(h.download_state()[0]))
    assert f.get_parameters(0)[0] == h.get_parameters(0)[0] != 0.6
    f.close()
    h.close()


# ---- a script-defined ACTION: multiplicative scaling x -> x exp(delta) ------------------------------------------------
# perform_action!: x exp(delta); invert_action!: -delta; delta ~ Normal(0, sigma).  The move's density in state space is
# q(x' | x) = N(delta; 0, sigma) / |x'|, so log_proposal_density carries -log|x exp(delta)| (its backward / forward
# difference is the Jacobian +delta).  Chains never change sign: started on x > 0 they sample the half-Gaussian
# p(x) ~ exp(-beta x^2), x > 0, whose <x^2> = 1/(2 beta) and <x> = 1/sqrt(pi beta).
SCALING = ("sigma*z",
           "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(sigma) - amc_log(fabs(x)) - delta",
           "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma",
           "x*amc_exp(delta)", "-delta")


def test_oracle_scaling_action_samples_the_half_gaussian(oracle):
    s = oracle.OracleSim(4000, potential="harmonic", beta=BETA, sigma=[0.8], weight=[1.0], seed=9, proposal=SCALING)
    s.set_x(np.random.default_rng(3).uniform(0.1, 2.0, 4000))
    n, sx, sxx, _ = s.run_pooled_moments(3000, 400, 10, threads=8)
    assert sxx / n == pytest.approx(0.25, abs=4e-3) and sx / n == pytest.approx(1 / np.sqrt(np.pi * BETA), abs=4e-3)
    # without the Jacobian term in logq the chain samples exp(-beta x^2) / x instead: <x^2> comes out far lower
    wrong = oracle.OracleSim(4000, potential="harmonic", beta=BETA, sigma=[0.8], weight=[1.0], seed=9,
                             proposal=(SCALING[0], "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(sigma)", None, SCALING[3], SCALING[4]))
    wrong.set_x(np.random.default_rng(3).uniform(0.1, 2.0, 4000))
    n, sx, sxx, _ = wrong.run_pooled_moments(3000, 400, 10, threads=8)
    assert sxx / n < 0.2
    oracle.install_custom_proposal(None)


def test_script_action_travels_through_the_host_mirror(oracle, tmp_path):
    chains = ma.ParticleChains(64, BETA, x=np.random.default_rng(5).uniform(0.2, 1.5, 64))
    move = ma.Move(ma.ScriptAction(perform=SCALING[3], invert=SCALING[4]), ma.ScriptPolicy(*SCALING[:3]), [0.5], 1.0)
    sim = ma.Simulation(chains, [dict(algorithm=ma.Metropolis, pool=[move], seed=8, engine_factory=oracle.OracleEngine)], 15,
                        path=str(tmp_path))
    ma.run(sim)
    o = oracle.OracleSim(64, potential="harmonic", beta=BETA, sigma=[0.5], weight=[1.0], seed=8, proposal=SCALING)
    o.set_x(np.random.default_rng(5).uniform(0.2, 1.5, 64))
    o.make_steps(15)
    assert np.array_equal(bits(chains.x), bits(o.state()[0])) and np.all(chains.x > 0)
    with pytest.raises(TypeError, match="needs a ScriptPolicy"):
        ma.Move(ma.ScriptAction(), ma.StandardGaussian(), [0.5], 1.0)
    oracle.install_custom_proposal(None)


def test_action_expressions_come_together(amc):
    kw = dict(n_chains=10, potential="harmonic", beta=BETA, sigma=[0.5], weight=[1.0])
    with pytest.raises(amc.AmcError, match="come together"):
        amc.HipEngine(proposal=(SCALING[0], SCALING[1], None, SCALING[3], None), **kw)
    with pytest.raises(amc.AmcError, match="does not mention delta"):
        amc.HipEngine(proposal=(SCALING[0], SCALING[1], None, "x*2.0", "-delta"), **kw)


@pytest.mark.gpu
def test_scaling_action_bit_exact_and_half_gaussian_on_the_device(gpu, oracle):
    M = 20001
    kw = dict(potential="harmonic", beta=BETA, sigma=[0.3, 0.9], weight=[0.5, 0.5], seed=14, proposal=SCALING)
    x0 = np.random.default_rng(6).uniform(0.1, 2.0, M)
    eng = gpu.HipEngine(n_chains=M, **kw)
    sim = oracle.OracleSim(M, **kw)
    eng.upload_state(x0)
    sim.set_x(x0)
    for n in (1, 1, 7, 20):
        eng.sweep(n)
        sim.make_steps(n)
        x, e = eng.download_state()
        xo, eo = sim.state()
        assert np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(e), bits(eo))
    acc, tot = eng.download_counters()
    ao, to = sim.counters()
    assert np.array_equal(acc, ao) and np.array_equal(tot, to)
    got, want = np.asarray(eng.pg_estimate([0, 1], 2)).reshape(2, 5), sim.pg_estimate([0, 1], 2)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)
    assert np.array_equal(bits(eng.download_state()[0]), bits(sim.state()[0]))         # revert = perform(x', invert(delta))
    eng.close()
    oracle.install_custom_proposal(None)
    big = gpu.HipEngine(n_chains=2_000_000, potential="harmonic", beta=BETA, sigma=[0.8], weight=[1.0], seed=15, proposal=SCALING,
                        per_chain_counters=False)
    big.upload_state(np.random.default_rng(7).uniform(0.1, 2.0, 2_000_000))
    big.sweep(500)
    s = np.zeros(4)
    for _ in range(30):
        big.sweep(20)
        s += big.reduce()[:4]
    assert s[2] / s[3] == pytest.approx(0.25, abs=2e-3) and s[1] / s[3] == pytest.approx(1 / np.sqrt(np.pi * BETA), abs=2e-3)
    big.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["proposal_k1", "proposal_k2", "scaled", "scaled_f32", "classes", "two_parameters"])
def test_accept_filter_equals_the_exact_decision_for_script_proposals(gpu, monkeypatch, case):
    """Round 5: the decisions of script-defined proposals go through the 12-bit accept filter too (accept_filter_arg: arg is
    formed in full, the filter saves exp(arg) and the accept draw).  2e6 chains x 300 steps = 6e8 decisions per case, once
    filtered and once with every wave sent through the reference-ordered decision (AMC_EXACT_ACCEPT=1): positions and
    counters must be identical -- single-step launches, the multi-step form and the fused sweep + estimator launch."""
    M = 2_000_000
    gauss = ("sigma*z", "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0",
             "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma")
    drift = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
             ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"])
    kw = dict(n_chains=M, potential="harmonic", beta=BETA, seed=23)
    learn = [0]
    if case == "proposal_k1":
        kw.update(sigma=[0.4], weight=[1.0], proposal=MALA, per_chain_counters=False)
    elif case == "proposal_k2":
        kw.update(potential="double_well", sigma=[0.2, 0.6], weight=[0.3, 0.7], proposal=MALA)
    elif case == "scaled":
        kw.update(sigma=[0.4], weight=[1.0], scale_expr="0.5 + fabs(x)")
    elif case == "scaled_f32":                       # Particle{Float32}: x, delta, dlogp in Float32, arg stays Float64
        kw.update(sigma=[0.4], weight=[1.0], scale_expr="0.5 + fabs(x)", dtype="f32")
    elif case == "classes":
        kw.update(sigma=[0.3, 0.4], weight=[0.5, 0.5], classes=[gauss, MALA], class_of_move=[0, 1])
    else:
        kw.update(sigma=[[0.05, 0.5]], weight=[1.0], proposal=drift, n_params=2)
    runs = []
    for exact in ("0", "1"):
        monkeypatch.setenv("AMC_EXACT_ACCEPT", exact)
        e = gpu.HipEngine(**kw)
        e.init_uniform(-2, 2)
        for _ in range(20):
            e.sweep(1)
        e.sweep(270)
        e.pgmc_steps(10, learn, 1, [1], [0.0], [0.0])            # fused sweep + estimator launches (eta = 0: nothing learns)
        acc, tot = e.counter_totals()
        runs.append((e.download_state(want_e=False)[0], np.asarray(acc), np.asarray(tot)))
        e.close()
    assert np.array_equal(bits(runs[0][0]), bits(runs[1][0]))
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    assert int(runs[0][2].sum()) == 300 * M and 0.2 < runs[0][1].sum() / runs[0][2].sum() < 0.99
