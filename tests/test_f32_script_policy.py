"""Float32 state under script-defined policies, actions and pools of classes.

The reference's Particle{T} / Displacement{T} are generic in T (example/particle_1d/particle_1d.jl:9,26) and so are the generic
functions a model defines for its policy (src/metropolis.jl:35-62): with T = Float32 the bodies run under Julia's promotion rules --
x and delta Float32, parameters and the normal variate Float64, Float32 x Float32 stays Float32, whatever meets a Float64 is
Float64, and a value assigned to a field of type T is converted.  C's usual arithmetic conversions say the same of the same text, so
the engine compiles the script's expressions with x and delta as floats (amc_model.h: real_t) and the oracle's twins do likewise
(tests/oracle_lib.py: the *_f32 functions; oracle/amc_oracle.c: mc_step_script_f32 / pgmc_sample_script_f32).  Everything below is
bit for bit."""
import numpy as np
import pytest

BETA = 2.0
GAUSS = ("sigma*z", "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0",
         "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma")
MALA = ("-2.0*sigma*sigma*x + sigma*z",
        "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)",
        "((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(sigma*sigma*sigma) - 4.0*x*(delta + 2.0*sigma*sigma*x)/sigma - 1.0/sigma")
SCALING = ("sigma*z", "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(sigma) - amc_log(fabs(x)) - delta",
           "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma", "x*amc_exp(delta)", "-delta")
DRIFT = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
         ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"])


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def is_f32(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array_equal(bits(a), bits(a.astype(np.float32).astype(np.float64)))


def auto(policy):
    """The policy without its hand-written derivative: the engine (and the twin) differentiate logq over dual numbers."""
    return (policy[0], policy[1], None) + tuple(policy[3:])


def _restore(oracle):
    oracle.install_custom_proposal(None)
    oracle.install_vector_policy(1, None)
    oracle.install_policy_classes(None, None)


def test_oracle_float32_twin_differs_from_the_float64_one_and_keeps_the_distribution(oracle):
    """No GPU: the Float32 twin takes its own roundings (positions are Float32 values and differ from the Float64 run's), and the
    sampler still samples exp(-beta x^2) (test/distribution_test.jl:33-37's statistic at its tolerance)."""
    kw = dict(potential="harmonic", beta=BETA, sigma=[0.5], weight=[1.0], seed=5, proposal=auto(MALA))
    a, b = oracle.OracleSim(3000, dtype="f32", **kw), None
    a.init_uniform(-2, 2)
    a.make_steps(5)
    x32 = a.state()[0]
    assert is_f32(x32)
    g = a.pg_estimate([0], 2)
    assert np.all(np.isfinite(g)) and g[0, 4] == 6000
    n, sx, sxx, _ = a.run_pooled_moments(2500, 300, 10, threads=8)
    assert sx / n == pytest.approx(0.0, abs=1e-2) and sxx / n == pytest.approx(1 / (2 * BETA), abs=5e-3)
    a.close()
    b = oracle.OracleSim(3000, dtype="f64", **kw)
    b.init_uniform(-2, 2)
    b.make_steps(5)
    assert not np.array_equal(x32, b.state()[0])
    b.close()
    _restore(oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mala_auto_k1", "gauss_hand_k2_beta", "width_custom_potential", "scaling_action"])
def test_one_parameter_policies_bit_exact(gpu, oracle, case):
    import montecarlo_amd as ma
    M = 20001
    kw = dict(potential="harmonic", beta=BETA, sigma=[0.4], weight=[1.0], seed=11, proposal=auto(MALA), dtype="f32")
    beta, lo = None, -2.0
    if case == "mala_auto_k1":
        kw["per_chain_counters"] = False
    elif case == "gauss_hand_k2_beta":
        kw.update(potential="double_well", sigma=[0.2, 0.6], weight=[0.4, 0.6], proposal=GAUSS)
        beta = np.random.default_rng(1).uniform(0.5, 3.0, M).astype(np.float32).astype(np.float64)
    elif case == "width_custom_potential":
        kw.update(potential=ma.CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"), reward_expr="fabs(delta)*(1.0 + x*x)",
                  proposal=("sigma*z*(0.5 + fabs(x))", "-(delta*delta)/(2.0*(sigma*(0.5 + fabs(x)))*(sigma*(0.5 + fabs(x)))) - amc_log(sigma*(0.5 + fabs(x)))", None))
    else:
        kw.update(proposal=auto(SCALING), sigma=[0.5, 0.2], weight=[0.5, 0.5])
        lo = 0.1
    eng, ref = gpu.HipEngine(n_chains=M, **kw), oracle.OracleEngine(n_chains=M, **kw)
    x0 = np.random.default_rng(2).uniform(lo, 2, M).astype(np.float32).astype(np.float64)
    for e in (eng, ref):
        e.upload_state(x0, beta)
    for n in (1, 1, 9, 30):
        eng.sweep(n)
        ref.sweep(n)
        x, e = eng.download_state()
        xo, eo = ref.download_state()
        assert is_f32(x) and is_f32(e)
        assert np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(e), bits(eo)), n
    acc, tot = eng.counter_totals()
    ao, to = ref.download_counters()
    assert np.array_equal(np.asarray(acc), ao.sum(axis=1)) and np.array_equal(np.asarray(tot), to.sum(axis=1))
    ids = list(range(len(kw["sigma"])))
    for q in (1, 3):
        assert np.array_equal(eng.pg_estimate_exact(ids, q), ref.pg_estimate_exact(ids, q), equal_nan=True), q
        assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    # a short device-resident PGMC run: sweep + estimator + learning step per launch, against the separate calls of the twin
    learn = [ids[-1]]
    eng.pgmc_steps(4, learn, 2, [1], [0.02], [0.0])
    for _ in range(4):
        ref.sweep(1)
        ref.pg_accumulate(learn, 2)
        ref.pg_update(learn, [1], [0.02], [0.0])
    assert eng.get_parameters(learn[0])[0] == ref.get_parameters(learn[0])[0] != kw["sigma"][-1]
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    eng.close()
    _restore(oracle)
    oracle.install_custom_reward(None)


@pytest.mark.gpu
@pytest.mark.parametrize("hand", [True, False], ids=["partials_given", "partials_differentiated"])
def test_two_parameter_policy_bit_exact(gpu, oracle, hand):
    M = 4099
    policy = DRIFT if hand else (DRIFT[0], DRIFT[1], None)
    kw = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[[0.1, 0.5], [-0.2, 1.1]], weight=[0.3, 0.7], seed=11, proposal=policy,
              n_params=2, dtype="f32")
    eng, ref = gpu.HipEngine(**kw), oracle.OracleEngine(**kw)
    for e in (eng, ref):
        e.init_uniform(-2.0, 2.0)
    for n in (1, 7, 40):
        eng.sweep(n)
        ref.sweep(n)
        x = eng.download_state()[0]
        assert is_f32(x) and np.array_equal(bits(x), bits(ref.download_state()[0])), n
    a, t = eng.download_counters()
    ao, to = ref.download_counters()
    assert np.array_equal(a, ao) and np.array_equal(t, to)
    for ids, q in (([1], 1), ([0, 1], 3)):
        assert np.array_equal(eng.pg_estimate_exact(ids, q), ref.pg_estimate_exact(ids, q), equal_nan=True), (ids, q)
    eng.pgmc_steps(5, [0, 1], 2, [1, 4], [0.02, 5e-3], [0.0, 1e-6])              # VPG and NPG
    for _ in range(5):
        ref.sweep(1)
        ref.pg_accumulate([0, 1], 2)
        ref.pg_update([0, 1], [1, 4], [0.02, 5e-3], [0.0, 1e-6])
    for k in (0, 1):
        assert np.array_equal(bits(eng.get_parameters(k)), bits(ref.get_parameters(k)))
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    eng.close()
    _restore(oracle)


@pytest.mark.gpu
def test_pool_of_classes_bit_exact(gpu, oracle):
    M = 4099
    kw = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.3, 0.6, 0.1, 0.5], weight=[0.3, 0.3, 0.2, 0.2], seed=21,
              classes=[GAUSS, auto(MALA), SCALING], class_of_move=[0, 1, 0, 2], dtype="f32")
    eng, ref = gpu.HipEngine(**kw), oracle.OracleEngine(**kw)
    for e in (eng, ref):
        e.init_uniform(0.1, 2.0)
    for n in (1, 6, 25):
        eng.sweep(n)
        ref.sweep(n)
        x = eng.download_state()[0]
        assert is_f32(x) and np.array_equal(bits(x), bits(ref.download_state()[0])), n
    a, t = eng.download_counters()
    ao, to = ref.download_counters()
    assert np.array_equal(a, ao) and np.array_equal(t, to)
    assert np.array_equal(eng.pg_estimate_exact([0, 1, 2, 3], 2), ref.pg_estimate_exact([0, 1, 2, 3], 2))
    eng.pgmc_steps(4, [1, 3], 2, [1, 2], [0.03, 0.02], [0.0, 0.0])
    for _ in range(4):
        ref.sweep(1)
        ref.pg_accumulate([1, 3], 2)
        ref.pg_update([1, 3], [1, 2], [0.03, 0.02], [0.0, 0.0])
    assert [eng.get_parameters(k)[0] for k in range(4)] == [ref.get_parameters(k)[0] for k in range(4)]
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    eng.close()
    _restore(oracle)
