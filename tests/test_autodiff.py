"""Derivatives the user does not have to write: d logq / d theta by forward-mode differentiation of the script's logq.

The reference never asks for a derivative: withgrad_log_proposal_density! evaluates the model's own log_proposal_density over
ForwardDiff's dual numbers (src/PolicyGuided/gradients.jl:28-33; ext/EnzymeExt.jl:8-19, ext/ZygoteExt.jl with their engines),
and test/ad_backends_test.jl:31-32 pins the three backends against each other to 1e-10.  Here amc_create_proposal_model /
_vector_policy_model / _mixed_model take NULL for the derivative expressions and the kernels evaluate AMC_USER_LOGQ over
Dual<P> (montecarlo_amd/csrc/amc_dual.h: ForwardDiff 0.10's rules, operation by operation); the oracle's twin has its own dual
numbers (tests/oracle_lib.py).  What is checked:
  * CPU: the models compile without derivative expressions (amc_model_check, no GPU); the oracle's dual-number gradient equals
    the hand-derived one to <= 4 ulp of the terms it adds, and test/ad_backends_test.jl's closed form (-5 at delta = 0,
    sigma = 0.2) to 1e-10;
  * GPU: GradientData records of a handle created WITHOUT dlogq equal the oracle's dual-number twin bit for bit (one parameter,
    two parameters, a pool of classes where only some classes bring their own expression); against the same policy WITH the
    hand-written expressions the records agree to rounding; free-running PGMC learns the same sigma to 1e-12 relative."""
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
# a policy that uses the rest of the vocabulary: sqrt, fabs, fma, amc_exp, a quotient of two parameter-dependent terms
WIDE = ("sigma*z*sqrt(1.0 + x*x)",
        "-(delta*delta)/(2.0*sigma*sigma*(1.0 + x*x)) - amc_log(sigma*sqrt(1.0 + x*x)) + fma(0.0, sigma, 0.0) + 0.0*amc_exp(-fabs(sigma))/(1.0 + sigma)",
        "(delta*delta)/(sigma*sigma*sigma*(1.0 + x*x)) - 1.0/sigma")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def auto(p):
    """The policy without its derivative expressions."""
    p = list(p)
    return tuple(p[:2] + [None] + p[3:])


# ---------------------------------------------------------------- CPU -------------------------------------------------
def test_models_without_derivative_expressions_compile(amc):
    assert amc.model_check(GAUSS[0], GAUSS[1]) == ""
    assert amc.model_check(MALA[0], MALA[1]) == ""
    assert amc.model_check(SCALING[0], SCALING[1], None, SCALING[3], SCALING[4]) == ""
    assert amc.model_check(DRIFT[0], DRIFT[1], None, n_params=2) == ""
    assert amc.model_check(WIDE[0], WIDE[1], potential="sqrt(fabs(x)) + fma(x, x, 0.5)") == ""
    # a pool of classes: some bring their own expression, some do not
    assert amc.model_check([GAUSS[0], MALA[0], SCALING[0]], [GAUSS[1], MALA[1], SCALING[1]], [GAUSS[2], None, None],
                           [None, None, SCALING[3]], [None, None, SCALING[4]]) == ""
    # what the dual numbers lack is a compile error with the compiler's own words, not a wrong number
    with pytest.raises(amc.AmcError, match="amc error -1"):
        amc.model_check("sigma*z", "-(delta*delta)/(2.0*sigma*sigma) - amc_log(sigma) + (sigma > 1.0 ? sigma : 1.0)")
    with pytest.raises(amc.AmcError, match="one expression per parameter, or none"):
        amc.model_check(DRIFT[0], DRIFT[1], [DRIFT[2][0], None], n_params=2)


def _oracle_dlogq(oracle, policy, delta, x, sigma):
    """d logq / d sigma of a one-parameter policy at (delta, x, sigma) through the oracle's compiled twin."""
    import ctypes as C
    oracle.install_custom_proposal(policy)
    import hashlib
    sample, logq, dlogq, perform, invert = (list(policy) + [None] * 4)[:5]
    key = "p" + hashlib.sha1(repr((sample, logq, dlogq, perform, invert)).encode()).hexdigest()[:16]
    fn = oracle._custom_libs[key].amo_user_dlogq
    fn.restype, fn.argtypes = C.c_double, [C.c_double] * 3
    out = np.array([fn(float(d), float(xx), float(s)) for d, xx, s in zip(delta, x, sigma)])
    oracle.install_custom_proposal(None)
    return out


def test_dual_number_gradient_equals_the_hand_derived_one(oracle):
    """The done-criterion of the round: hand-derived and dual-number gradients agree to <= 4 ulp -- of the magnitudes the formulas
    add up (a difference of two terms can cancel: the ulp is that of the larger term, as for any two orderings of one sum)."""
    rng = np.random.default_rng(5)
    n = 4000
    sigma = np.exp(rng.uniform(np.log(0.05), np.log(5.0), n))
    x = rng.normal(0, 1.0, n)
    z = rng.normal(0, 1.0, n)
    cases = {"gauss": (GAUSS, sigma * z, lambda d, xx, s: np.abs(d * d / s ** 3) + 1 / s),
             "mala": (MALA, -2 * sigma * sigma * x + sigma * z,
                      lambda d, xx, s: (d + 2 * s * s * xx) ** 2 / s ** 3 + np.abs(4 * xx * (d + 2 * s * s * xx) / s) + 1 / s),
             "scaling": (SCALING, sigma * z, lambda d, xx, s: d * d / s ** 3 + 1 / s),
             "wide": (WIDE, sigma * z * np.sqrt(1 + x * x), lambda d, xx, s: d * d / (s ** 3 * (1 + xx * xx)) + 1 / s + 1.0)}
    for name, (pol, delta, scale) in cases.items():
        hand = _oracle_dlogq(oracle, pol, delta, x, sigma)
        dual = _oracle_dlogq(oracle, auto(pol), delta, x, sigma)
        assert np.all(np.isfinite(dual)), name
        assert np.all(np.abs(dual - hand) <= 4 * 2.0 ** -52 * scale(delta, x, sigma)), (name, np.max(np.abs(dual - hand) / scale(delta, x, sigma)) / 2.0 ** -52)
    # test/ad_backends_test.jl:27-32: delta = 0, sigma = 0.2 -> -5.0 (atol 1e-10), the policy of the reference's own example
    at = _oracle_dlogq(oracle, auto(GAUSS), [0.0], [0.3], [0.2])[0]
    assert abs(at - (-5.0)) < 1e-10


def test_oracle_two_parameter_dual_gradient(oracle):
    import ctypes as C
    import hashlib
    rng = np.random.default_rng(6)
    th = np.stack([rng.normal(0, 0.3, 500), np.exp(rng.uniform(np.log(0.1), np.log(3), 500))], axis=1)
    x, z = rng.normal(0, 1, 500), rng.normal(0, 1, 500)
    delta = th[:, 0] + th[:, 1] * z
    outs = []
    for pol in ((DRIFT[0], DRIFT[1], DRIFT[2]), (DRIFT[0], DRIFT[1], None)):
        oracle.install_vector_policy(2, pol)
        key = "v" + hashlib.sha1(repr((2, pol[0], pol[1], pol[2], None, None)).encode()).hexdigest()[:16]
        fn = oracle._custom_libs[key].amo_vec_dlogq
        fn.restype, fn.argtypes = None, [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        out = np.zeros((500, 2))
        for i in range(500):
            t = (C.c_double * 2)(*th[i])
            o = (C.c_double * 2)()
            fn(delta[i], x[i], t, o)
            out[i] = o[0], o[1]
        outs.append(out)
    oracle.install_vector_policy(1, None)
    hand, dual = outs
    scale = np.stack([np.abs(delta - th[:, 0]) / th[:, 1] ** 2, (delta - th[:, 0]) ** 2 / th[:, 1] ** 3 + 1 / th[:, 1]], axis=1)
    assert np.all(np.abs(dual - hand) <= 4 * 2.0 ** -52 * np.maximum(scale, 1e-300))


# ---------------------------------------------------------------- GPU -------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,policy", [("gauss", GAUSS), ("mala", MALA), ("scaling", SCALING), ("wide", WIDE)])
def test_records_without_dlogq_equal_the_oracles_dual_number_twin(gpu, oracle, name, policy):
    """A handle created WITHOUT the derivative runs the estimator, and its GradientData records equal the oracle's -- whose twin
    differentiates the same text with its own dual numbers -- bit for bit; positions too.  Against the hand-written
    derivative: the same j, gradients equal to rounding."""
    M = 4099
    kw = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.6, 0.3], weight=[0.5, 0.5], seed=31)
    a = gpu.HipEngine(proposal=auto(policy), **kw)
    o = oracle.OracleEngine(proposal=auto(policy), **kw)
    h = gpu.HipEngine(proposal=policy, **kw)
    lo, hi = (0.1, 2.0) if name == "scaling" else (-2.0, 2.0)
    for e in (a, o, h):
        e.init_uniform(lo, hi)
        e.sweep(7)
    ra, ro, rh = a.pg_estimate_exact([0, 1], 3), o.pg_estimate_exact([0, 1], 3), h.pg_estimate_exact([0, 1], 3)
    assert np.array_equal(ra, ro)
    assert np.array_equal(bits(a.download_state()[0]), bits(o.download_state()[0]))
    va, vh = gpu.xsum_round(ra).reshape(2, 5), gpu.xsum_round(rh).reshape(2, 5)
    assert np.array_equal(va[:, 0], vh[:, 0]) and np.array_equal(va[:, 4], vh[:, 4])          # j and n do not involve the derivative
    assert np.allclose(va[:, 1:4], vh[:, 1:4], rtol=1e-11, atol=1e-9 * M)
    # the fused time step (sweep + estimator + learning step in one launch) with the differentiated density: equal to the oracle
    a.pgmc_steps(5, [0, 1], 2, [1, 2], [0.02, 0.01], [0.0, 0.0])
    for _ in range(5):
        o.sweep(1)
        o.pg_accumulate([0, 1], 2)
        o.pg_update([0, 1], [1, 2], [0.02, 0.01], [0.0, 0.0])
    assert [a.get_parameters(k)[0] for k in range(2)] == [o.get_parameters(k)[0] for k in range(2)]
    assert np.array_equal(bits(a.download_state()[0]), bits(o.download_state()[0]))
    for e in (a, h):
        e.close()
    oracle.install_custom_proposal(None)


@pytest.mark.gpu
def test_two_parameter_policy_without_partials(gpu, oracle):
    M = 6001
    kw = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[[0.05, 0.5], [-0.1, 0.9]], weight=[0.4, 0.6], seed=33, n_params=2)
    a = gpu.HipEngine(proposal=(DRIFT[0], DRIFT[1], None), **kw)
    o = oracle.OracleEngine(proposal=(DRIFT[0], DRIFT[1], None), **kw)
    h = gpu.HipEngine(proposal=DRIFT, **kw)
    for e in (a, o, h):
        e.init_uniform(-2.0, 2.0)
        e.sweep(5)
    ra, ro, rh = a.pg_estimate_exact([0, 1], 2), o.pg_estimate_exact([0, 1], 2), h.pg_estimate_exact([0, 1], 2)
    assert ra.shape == ro.shape and np.array_equal(ra, ro)
    va, vh = gpu.xsum_round(ra).reshape(2, -1), gpu.xsum_round(rh).reshape(2, -1)
    assert np.array_equal(va[:, 0], vh[:, 0]) and np.allclose(va, vh, rtol=1e-11, atol=1e-9 * M)
    # free-running PGMC with the natural-gradient optimiser (2 x 2 metric inverted on the device): equal to the oracle's run
    a.pgmc_steps(6, [1], 2, [6], [1e-4], [1e-6])              # BLANPG
    for _ in range(6):
        o.sweep(1)
        o.pg_accumulate([1], 2)
        o.pg_update([1], [6], [1e-4], [1e-6])
    assert np.array_equal(a.get_parameters(1), o.get_parameters(1)) and not np.array_equal(a.get_parameters(1), [-0.1, 0.9])
    assert np.array_equal(bits(a.download_state()[0]), bits(o.download_state()[0]))
    for e in (a, h):
        e.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
def test_pool_of_classes_where_only_some_bring_a_derivative(gpu, oracle):
    """Gaussian class with its canonical expressions (table-row shortcut on), Langevin and scaling classes differentiated by the
    engine: all four moves learn; records and a short device-resident run equal the oracle's."""
    M = 4099
    classes = [GAUSS, auto(MALA), auto(SCALING)]
    kw = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.3, 0.6, 0.1, 0.5], weight=[0.3, 0.3, 0.2, 0.2], seed=21,
              classes=classes, class_of_move=[0, 1, 0, 2])
    a, o = gpu.HipEngine(**kw), oracle.OracleEngine(**kw)
    for e in (a, o):
        e.init_uniform(0.1, 2.0)
        e.sweep(6)
    assert np.array_equal(a.pg_estimate_exact([0, 1, 2, 3], 2), o.pg_estimate_exact([0, 1, 2, 3], 2))
    a.pgmc_steps(4, [1, 3], 2, [1, 2], [0.03, 0.02], [0.0, 0.0])
    for _ in range(4):
        o.sweep(1)
        o.pg_accumulate([1, 3], 2)
        o.pg_update([1, 3], [1, 2], [0.03, 0.02], [0.0, 0.0])
    assert [a.get_parameters(k)[0] for k in range(4)] == [o.get_parameters(k)[0] for k in range(4)]
    assert np.array_equal(bits(a.download_state()[0]), bits(o.download_state()[0]))
    a.close()
    # the other way round: class 0 (the unsuffixed expressions) differentiated, class 1 with its own derivative
    kw2 = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.3, 0.6], weight=[0.5, 0.5], seed=22,
               classes=[auto(GAUSS), MALA], class_of_move=[0, 1])
    a, o = gpu.HipEngine(**kw2), oracle.OracleEngine(**kw2)
    for e in (a, o):
        e.init_uniform(-2.0, 2.0)
        e.sweep(6)
    assert np.array_equal(a.pg_estimate_exact([0, 1], 2), o.pg_estimate_exact([0, 1], 2))
    assert a.pg_route(2, 2, fused=True)[0]
    a.pgmc_steps(3, [0, 1], 2, [1, 1], [0.03, 0.02], [0.0, 0.0])
    for _ in range(3):
        o.sweep(1)
        o.pg_accumulate([0, 1], 2)
        o.pg_update([0, 1], [1, 1], [0.03, 0.02], [0.0, 0.0])
    assert [a.get_parameters(k)[0] for k in range(2)] == [o.get_parameters(k)[0] for k in range(2)]
    assert np.array_equal(bits(a.download_state()[0]), bits(o.download_state()[0]))
    a.close()
    oracle.install_policy_classes(None, None)


@pytest.mark.gpu
def test_host_mirror_learns_with_a_policy_that_has_no_derivative(gpu, oracle, tmp_path):
    """Through Simulation / run with ScriptPolicy(sample, logq) -- no dlogq -- as a user of the reference would write it: the
    learned sigma equals the run with the hand-written derivative to 1e-9 relative (two orderings of the same sums)."""
    import montecarlo_amd as ma
    out = []
    for pol in (ma.ScriptPolicy(MALA[0], MALA[1]), ma.ScriptPolicy(*MALA)):
        chains = ma.ParticleChains.uniform(20001, BETA, -2.0, 2.0)
        pool = [ma.Move(ma.Displacement(0.0), pol, [0.3], 1.0)]
        algos = [dict(algorithm=ma.Metropolis, pool=pool, seed=5),
                 dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=[ma.VPG(0.02)], q_batch_size=2),
                 dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,))]
        sim = ma.Simulation(chains, algos, 40, path=str(tmp_path / str(len(out))))
        ma.run(sim)
        out.append(pool[0].parameters[0])
    assert out[0] != 0.3 and out[0] == pytest.approx(out[1], rel=1e-9)
