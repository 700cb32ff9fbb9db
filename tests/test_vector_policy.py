"""Policies with SEVERAL parameters (amc_create_vector_policy_model; HipEngine(n_params=P)).

Move.parameters is an array in the reference (src/metropolis.jl:140-147), GradientData keeps grad j and grad logq_forward as
arrays of its shape and g as their outer product (src/PolicyGuided/gradients.jl:41-61,104-108), and the natural-gradient
optimisers invert g + eps I (learning.jl:103-104,130-133,159-163).  The example policies:

  * DRIFT (P = 2): delta = theta0 + theta1 z -- a Gaussian displacement with a learnable mean, the first thing a user of
    this path writes; the proposal is no longer symmetric, so log q_backward - log q_forward = -2 delta theta0 / theta1^2
    enters every acceptance;
  * LEAN (P = 3): delta ~ Normal(theta0 + theta2 x, theta1) -- the mean leans on the state, so forward and backward
    densities (and their gradients) are taken at different states (gradients.jl:97,102).

The oracle evaluates the same expressions compiled by gcc (tests/oracle_lib.py install_vector_policy)."""
import numpy as np
import pytest

BETA = 2.0
DRIFT = ("theta0 + theta1*z",
         "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
         ["(delta-theta0)/(theta1*theta1)",
          "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"])
_R = "(delta-theta0-theta2*x)"
LEAN = (f"theta0 + theta2*x + theta1*z",
        f"-({_R}*{_R})/(2.0*theta1*theta1) - amc_log(theta1)",
        [f"{_R}/(theta1*theta1)", f"({_R}*{_R})/(theta1*theta1*theta1) - 1.0/theta1", f"{_R}*x/(theta1*theta1)"])
OPTS = [("VPG", 1, 2e-2, 0.0), ("BLPG", 2, 2e-2, 0.0), ("BLAPG", 3, 1e-5, 1e-6), ("NPG", 4, 5e-3, 1e-6),
        ("ANPG", 5, 1e-5, 1e-6), ("BLANPG", 6, 1e-5, 1e-6)]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# ---- CPU: the oracle's own pieces ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_small_inverse_against_numpy(oracle, P):
    import ctypes as C
    lib = oracle.load()
    lib.amo_inv_small.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
    lib.amo_inv_small.restype = C.c_int
    rng = np.random.default_rng(P)
    for _ in range(50):
        v = rng.normal(size=(8, P))
        A = v.T @ v / 8 + 1e-3 * np.eye(P)            # what g + eps I looks like: symmetric positive definite
        out = np.zeros((P, P))
        assert lib.amo_inv_small(oracle._dptr(np.ascontiguousarray(A)), P, oracle._dptr(out)) == 1
        assert np.allclose(out, np.linalg.inv(A), rtol=1e-9, atol=0)
    # a matrix that needs its rows swapped, and a singular one
    A = np.array([[0.0, 2.0], [3.0, 1.0]])
    out = np.zeros((2, 2))
    assert lib.amo_inv_small(oracle._dptr(A), 2, oracle._dptr(out)) == 1 and np.allclose(out @ A, np.eye(2), atol=1e-15)
    assert lib.amo_inv_small(oracle._dptr(np.array([[1.0, 2.0], [2.0, 4.0]])), 2, oracle._dptr(out)) == 0


@pytest.mark.parametrize("opt", OPTS, ids=[o[0] for o in OPTS])
def test_learning_steps_against_the_reference_formulas(oracle, opt):
    """learning.jl:32-34,50-52,77-79,103-105,130-134,160-164 written with numpy arrays, as the reference writes them."""
    name, _, h0, h1 = opt
    rng = np.random.default_rng(7)
    for P in (1, 2, 3, 4):
        v = rng.normal(size=(16, P))
        g = v.T @ v / 16
        j, dj, dl = 0.3, rng.normal(size=P) * 0.1, rng.normal(size=P) * 0.2
        theta = rng.normal(size=P)
        gd = np.concatenate([[j], dj, dl, g.reshape(-1)])
        got = oracle.learning_step_vec(name, h0, h1, theta, gd)
        F = g + h1 * np.eye(P)
        bj = dj - j * dl
        want = {"VPG": lambda: theta + h0 * dj,
                "BLPG": lambda: theta + h0 * bj,
                "BLAPG": lambda: theta + np.sqrt(2 * h0 / (dj @ dj + h1)) * bj,
                "NPG": lambda: theta + h0 * np.linalg.inv(F) @ dj,
                "ANPG": lambda: theta + np.sqrt(2 * h0 / (dj @ (np.linalg.inv(F) @ dj))) * np.linalg.inv(F) @ dj,
                "BLANPG": lambda: theta + np.sqrt(2 * h0 / (bj @ (np.linalg.inv(F) @ bj))) * np.linalg.inv(F) @ bj}[name]()
        assert np.allclose(got, want, rtol=1e-9, atol=0), (name, P)
        if P == 1:      # the one-parameter form, bit for bit
            assert got[0] == oracle.learning_step(name, h0, h1, float(theta[0]), [j, dj[0], dl[0], g[0, 0]])


def test_oracle_keeps_the_target_distribution_with_a_drifting_proposal(oracle):
    s = oracle.OracleSim(4000, potential="harmonic", beta=BETA, sigma=[[0.15, 0.6]], weight=[1.0], seed=5, proposal=DRIFT, n_params=2)
    s.init_uniform(-2, 2)
    n, sx, sxx, _ = s.run_pooled_moments(2500, 300, 10, threads=8)
    assert sx / n == pytest.approx(0.0, abs=6e-3) and sxx / n == pytest.approx(1 / (2 * BETA), abs=4e-3)
    oracle.install_vector_policy(1, None)


def test_argument_validation_needs_no_gpu(amc):
    kw = dict(n_chains=10, potential="harmonic", beta=BETA, weight=[1.0])
    with pytest.raises(amc.AmcError, match="needs a script-defined proposal"):
        amc.HipEngine(sigma=[[0.1, 0.5]], n_params=2, **kw)
    with pytest.raises(amc.AmcError, match="one vector of 2 parameters"):
        amc.HipEngine(sigma=[[0.1, 0.5, 0.2]], n_params=2, proposal=DRIFT, **kw)
    with pytest.raises(amc.AmcError, match="must list the 2 partial"):
        amc.HipEngine(sigma=[[0.1, 0.5]], n_params=2, proposal=(DRIFT[0], DRIFT[1], DRIFT[2][:1]), **kw)
    with pytest.raises(amc.AmcError, match=r"n_params must be in \[1, 4\]"):
        amc.HipEngine(sigma=[[0.1] * 5], n_params=5, proposal=(DRIFT[0], DRIFT[1], None), **kw)
    with pytest.raises(amc.AmcError, match="does not mention z"):
        amc.HipEngine(sigma=[[0.1, 0.5]], n_params=2, proposal=("theta0 + theta1", DRIFT[1], None), **kw)


# ---- GPU: parity with the oracle ----------------------------------------------------------------------------------------

def _pair(gpu, oracle, M, policy, P, thetas, weight, **kw):
    args = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=thetas, weight=weight, seed=11, proposal=policy, n_params=P, **kw)
    eng, ref = gpu.HipEngine(device=0, **args), oracle.OracleEngine(**args)
    eng.init_uniform(-2.0, 2.0)
    ref.init_uniform(-2.0, 2.0)
    return eng, ref


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["drift_k1", "drift_k2", "lean_k2"])
def test_sweeps_are_bit_exact(gpu, oracle, case):
    policy, P = (LEAN, 3) if case.startswith("lean") else (DRIFT, 2)
    if case == "drift_k1":
        thetas, weight = [[0.1, 0.5]], [1.0]
    elif case == "drift_k2":
        thetas, weight = [[0.1, 0.5], [-0.2, 1.1]], [0.3, 0.7]
    else:
        thetas, weight = [[0.05, 0.4, -0.3], [0.0, 1.0, 0.1]], [0.5, 0.5]
    eng, ref = _pair(gpu, oracle, 4099, policy, P, thetas, weight)
    assert np.array_equal(eng.get_parameters(len(thetas) - 1), np.array(thetas[-1]))
    for n in (1, 7, 40):
        eng.sweep(n)
        ref.sweep(n)
        assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0])), n
    a, t = eng.download_counters()
    ao, to = ref.download_counters()
    assert np.array_equal(a, ao) and np.array_equal(t, to)
    assert 0.2 < a.sum() / t.sum() < 0.95
    eng.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
@pytest.mark.parametrize("case,M,q_batch", [("drift", 1, 1), ("drift", 4099, 3), ("drift", 100003, 1), ("lean", 4099, 2), ("lean", 513, 40)])
def test_gradient_data_records_equal_the_oracles(gpu, oracle, case, M, q_batch):
    """j, grad j [P], grad logq [P] and the P x P g of two learnable moves as records: EQUAL to the oracle's (integer sums)."""
    policy, P, thetas = (LEAN, 3, [[0.05, 0.4, -0.3], [0.0, 1.0, 0.1]]) if case == "lean" else (DRIFT, 2, [[0.1, 0.5], [-0.2, 1.1]])
    eng, ref = _pair(gpu, oracle, M, policy, P, thetas, [0.4, 0.6])
    eng.sweep(3)
    ref.sweep(3)
    stride = 2 + 2 * P + P * P
    for ids in ([1], [0, 1]):
        got, want = eng.pg_estimate_exact(ids, q_batch), ref.pg_estimate_exact(ids, q_batch)
        assert got.shape == want.shape == (len(ids), stride, 12)
        assert np.array_equal(got, want, equal_nan=True), (case, ids)
    vals = eng.pg_estimate([0, 1], q_batch)
    assert np.array_equal(bits(vals), bits(ref.pg_estimate([0, 1], q_batch)))
    g = vals[0, 1 + 2 * P:1 + 2 * P + P * P].reshape(P, P)
    assert np.array_equal(g, g.T) and vals[0, -1] == M * q_batch
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))      # x = (x + d) + (-d) per sample
    assert eng.estimator_step == ref.estimator_step == 3
    eng.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", OPTS, ids=[o[0] for o in OPTS])
def test_free_running_pgmc_every_optimiser(gpu, oracle, opt):
    """[Metropolis, estimator, update] time steps on the device and in the oracle, nothing fed back: the parameter VECTORS are
    equal after every update, for each learning_step! -- the natural-gradient ones through the same 2 x 2 elimination."""
    name, kind, h0, h1 = opt
    eng, ref = _pair(gpu, oracle, 20011, DRIFT, 2, [[0.1, 0.3], [0.0, 0.8]], [0.5, 0.5])
    ids = [0, 1]
    for stretch in (1, 2, 12):
        eng.pgmc_steps(stretch, ids, 2, [kind, kind], [h0, h0], [h1, h1])
        ref.pgmc_steps(stretch, ids, 2, [kind, kind], [h0, h0], [h1, h1])
        for k in ids:
            assert np.array_equal(bits(eng.get_parameters(k)), bits(ref.get_parameters(k))), (name, stretch, k)
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    assert not np.array_equal(eng.get_parameters(0), [0.1, 0.3])            # it moved
    assert np.all(eng.pg_get_accumulated(ids) == 0.0)                       # initialise_gradient_data after the update
    # separate calls, accumulators kept across two estimator calls, then one update
    for e in (eng, ref):
        e.sweep(1)
        e.pg_accumulate(ids, 3)
        e.sweep(1)
        e.pg_accumulate(ids, 3)
    acc, acc_o = eng.pg_get_accumulated(ids), ref.pg_get_accumulated(ids)
    assert acc.shape == (2, 10) and np.array_equal(bits(acc), bits(acc_o)) and acc[0, -1] == 2 * 3 * 20011
    eng.pg_set_accumulated(ids, acc)                                         # resume path: what was read can be put back
    for e in (eng, ref):
        e.pg_update(ids, [kind, kind], [h0, h0], [h1, h1])
    for k in ids:
        assert np.array_equal(bits(eng.get_parameters(k)), bits(ref.get_parameters(k)))
    eng.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
def test_a_step_that_cannot_be_taken_is_not_applied(gpu):
    """NPG with eps = 0 on an exactly singular g (Julia's inv throws SingularException there; the reference would stop)."""
    eng = gpu.HipEngine(n_chains=8, device=0, potential="harmonic", beta=BETA, sigma=[[0.1, 0.5]], weight=[1.0], seed=2,
                        proposal=DRIFT, n_params=2)
    eng.init_uniform(-1, 1)
    #                 j    grad j     grad logq   g (singular)          n
    eng.pg_set_accumulated([0], np.array([[0.5, 0.1, -0.2, 0.0, 0.0, 1.0, 2.0, 2.0, 4.0, 1.0]]))
    eng.pg_update([0], [4], [1e-2], [0.0])
    assert np.array_equal(eng.get_parameters(0), [0.1, 0.5])
    with pytest.raises(gpu.AmcError, match="singular metric"):
        eng.pg_get_accumulated([0])
    # with eps > 0 the same data give a step
    eng2 = gpu.HipEngine(n_chains=8, device=0, potential="harmonic", beta=BETA, sigma=[[0.1, 0.5]], weight=[1.0], seed=2,
                         proposal=DRIFT, n_params=2)
    eng2.pg_set_accumulated([0], np.array([[0.5, 0.1, -0.2, 0.0, 0.0, 1.0, 2.0, 2.0, 4.0, 1.0]]))
    eng2.pg_update([0], [4], [1e-2], [1e-3])
    F = np.array([[1.0, 2.0], [2.0, 4.0]]) + 1e-3 * np.eye(2)
    assert np.allclose(eng2.get_parameters(0), np.array([0.1, 0.5]) + 1e-2 * np.linalg.inv(F) @ np.array([0.1, -0.2]), rtol=1e-9)
    eng.close()
    eng2.close()


@pytest.mark.gpu
def test_sums_do_not_depend_on_the_split_into_shards(gpu, oracle):
    kw = dict(n_chains=30011, potential="harmonic", beta=BETA, sigma=[[0.1, 0.5], [-0.2, 1.1]], weight=[0.4, 0.6], seed=9,
              proposal=DRIFT, n_params=2)
    one, three = gpu.HipEngine(device=0, **kw), gpu.SplitEngine(device=0, n_parts=3, **kw)
    for e in (one, three):
        e.init_uniform(-2, 2)
        e.sweep(4)
    assert np.array_equal(one.pg_estimate_exact([0, 1], 2), three.pg_estimate_exact([0, 1], 2))
    one.close()
    three.close()


@pytest.mark.gpu
def test_theta0_is_another_name_of_sigma(gpu):
    """One parameter: the expressions may say theta0 where the one-parameter entry points say sigma -- same kernels, same bits."""
    mala = ("-2.0*sigma*sigma*x + sigma*z",
            "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)",
            "((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(sigma*sigma*sigma) - 4.0*x*(delta + 2.0*sigma*sigma*x)/sigma - 1.0/sigma")
    kw = dict(n_chains=5001, device=0, potential="harmonic", beta=BETA, sigma=[0.4], weight=[1.0], seed=4)
    a = gpu.HipEngine(proposal=mala, **kw)
    b = gpu.HipEngine(proposal=(mala[0].replace("sigma", "theta0"), mala[1].replace("sigma", "theta0"), mala[2].replace("sigma", "theta0")), **kw)
    for e in (a, b):
        e.init_uniform(-2, 2)
        e.sweep(5)
    assert np.array_equal(a.pg_estimate_exact([0], 2), b.pg_estimate_exact([0], 2))
    assert np.array_equal(bits(a.download_state()[0]), bits(b.download_state()[0]))
    a.close()
    b.close()


# ---- the host mirror ------------------------------------------------------------------------------------------------------

def _pgmc_simulation(ma, engine_factory, path, device_resident, steps=24):
    chains = ma.ParticleChains.uniform(3001, BETA, -2.0, 2.0)
    pol = ma.ScriptPolicy(DRIFT[0], DRIFT[1], DRIFT[2], n_params=2)
    pool = [ma.Move(ma.Displacement(0.0), pol, [0.1, 0.3], 0.5), ma.Move(ma.Displacement(0.0), pol, [0.0, 0.8], 0.5)]
    opts = [ma.Static(), ma.BLANPG(1e-5, 1e-6)]
    algos = [dict(algorithm=ma.Metropolis, pool=pool, seed=3, engine_factory=engine_factory),
             dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=opts, q_batch_size=2,
                  device_resident=device_resident),
             dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,), scheduler=list(range(0, steps + 1, 3))),
             dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=list(range(0, steps + 1, 6)))]
    sim = ma.Simulation(chains, algos, steps, path=str(path))
    ma.run(sim)
    return chains, pool


def test_host_mirror_runs_a_two_parameter_policy(oracle, tmp_path):
    """The reference's own arrangement -- [Metropolis, PolicyGradientEstimator, PolicyGradientUpdate] with a ScriptPolicy of
    two parameters -- through the host mirror on the CPU test double: the host path (GradientData folded and learning_step!
    taken on the host, numpy) and the engine-resident path agree to rounding (numpy's inv against the engine's elimination)."""
    import montecarlo_amd as ma
    with pytest.raises(ValueError, match="this policy has 2 parameters"):
        ma.Move(ma.Displacement(0.0), ma.ScriptPolicy(DRIFT[0], DRIFT[1], DRIFT[2], n_params=2), [0.3], 1.0)
    with pytest.raises(ValueError, match="must list the 2 partial"):
        ma.ScriptPolicy(DRIFT[0], DRIFT[1], DRIFT[2][0], n_params=2)
    ch_h, pool_h = _pgmc_simulation(ma, oracle.OracleEngine, tmp_path / "host", False)
    ch_d, pool_d = _pgmc_simulation(ma, oracle.OracleEngine, tmp_path / "dev", True)
    assert pool_h[0].parameters.tolist() == [0.1, 0.3]                      # Static
    assert not np.array_equal(pool_h[1].parameters, [0.0, 0.8])
    assert np.allclose(pool_h[1].parameters, pool_d[1].parameters, rtol=1e-9, atol=1e-12)
    # StoreParameters writes "$(t) $(collect(parameters))" (src/metropolis.jl:438-443): "24 [a, b]"
    lines = (tmp_path / "dev" / "parameters" / "2" / "parameters.dat").read_text().splitlines()
    assert len(lines) == 5 and lines[0] == "0 [0.0, 0.8]" and lines[-1].startswith("24 [")
    last = [float(v) for v in lines[-1].split(" ", 1)[1].strip("[]").split(",")]
    assert last == pool_d[1].parameters.tolist()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
def test_host_mirror_on_the_engine_equals_the_test_double(gpu, oracle, tmp_path):
    import montecarlo_amd as ma
    ch_g, pool_g = _pgmc_simulation(ma, None, tmp_path / "gpu", True)
    ch_o, pool_o = _pgmc_simulation(ma, oracle.OracleEngine, tmp_path / "cpu", True)
    assert np.array_equal(bits(pool_g[1].parameters), bits(pool_o[1].parameters))
    assert np.array_equal(bits(ch_g.x), bits(ch_o.x))
    oracle.install_vector_policy(1, None)


# ---- three and four parameters: the 3 x 3 and 4 x 4 metric inverted on the device ------------------------------------------
_R4 = "(delta-theta0-theta2*x-theta3*x*x)"
BEND = ("theta0 + theta2*x + theta3*x*x + theta1*z",
        f"-({_R4}*{_R4})/(2.0*theta1*theta1) - amc_log(theta1)",
        [f"{_R4}/(theta1*theta1)", f"({_R4}*{_R4})/(theta1*theta1*theta1) - 1.0/theta1", f"{_R4}*x/(theta1*theta1)",
         f"{_R4}*x*x/(theta1*theta1)"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["lean3", "bend4"])
@pytest.mark.parametrize("opt", [OPTS[3], OPTS[5]], ids=["NPG", "BLANPG"])
def test_free_running_pgmc_with_three_and_four_parameters(gpu, oracle, case, opt):
    """The natural-gradient steps with a 3 x 3 and a 4 x 4 metric (Gauss-Jordan with partial pivoting on the device, the same
    sequence in the oracle): parameter vectors equal after every update, nothing fed back."""
    name, kind, h0, h1 = opt
    policy, P, thetas = (LEAN, 3, [[0.05, 0.4, -0.3], [0.0, 0.9, 0.1]]) if case == "lean3" else \
                        (BEND, 4, [[0.05, 0.4, -0.3, 0.02], [0.0, 0.9, 0.1, -0.05]])
    eng, ref = _pair(gpu, oracle, 20011, policy, P, thetas, [0.5, 0.5])
    ids = [0, 1]
    for stretch in (1, 2, 9):
        eng.pgmc_steps(stretch, ids, 2, [kind, kind], [h0, h0], [h1, h1])
        ref.pgmc_steps(stretch, ids, 2, [kind, kind], [h0, h0], [h1, h1])
        for k in ids:
            assert np.array_equal(bits(eng.get_parameters(k)), bits(ref.get_parameters(k))), (name, case, stretch, k)
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    assert not np.array_equal(eng.get_parameters(0), thetas[0]) and np.all(np.isfinite(eng.get_parameters(0)))
    acc = eng.pg_get_accumulated(ids)
    assert acc.shape == (2, 2 + 2 * P + P * P) and np.all(acc == 0.0)
    eng.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", OPTS, ids=[o[0] for o in OPTS])
@pytest.mark.parametrize("shape", ["K1", "K2"])
def test_one_learnable_move_takes_the_single_launch_time_step(gpu, oracle, opt, shape):
    """ONE move of a several-parameter policy learns: sweep + estimator + gradients_data += + learning step are ONE launch per
    time step (round 5; the tail is generic in P), where two learnable moves take a launch each (round 6: with their tails in them).  Same
    bits as the free-running oracle after every stretch -- parameter vectors, positions, counters, the callback sums that ride
    in the last launch, and the accumulators when no update follows."""
    name, kind, h0, h1 = opt
    if shape == "K1":
        eng, ref = _pair(gpu, oracle, 20011, DRIFT, 2, [[0.05, 0.5]], [1.0])
        ids = [0]
    else:
        eng, ref = _pair(gpu, oracle, 20011, DRIFT, 2, [[0.1, 0.3], [0.0, 0.8]], [0.5, 0.5])
        ids = [1]
    for stretch in (1, 3, 9):
        eng.pgmc_steps(stretch, ids, 2, [kind], [h0], [h1])
        ref.pgmc_steps(stretch, ids, 2, [kind], [h0], [h1])
        for k in range(eng.n_moves):
            assert np.array_equal(bits(eng.get_parameters(k)), bits(ref.get_parameters(k))), (name, stretch, k)
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    a, t = eng.download_counters()
    ao, to = ref.download_counters()
    assert np.array_equal(a, ao) and np.array_equal(t, to)
    assert np.all(eng.pg_get_accumulated(ids) == 0.0)
    # the callback sums of the state the last step leaves, formed in its launch
    eng.pgmc_steps(2, ids, 1, [kind], [h0], [h1], reduce_begin=True)
    ref.pgmc_steps(2, ids, 1, [kind], [h0], [h1], reduce_begin=True)
    ra, sa = eng.reduce_end_exact()
    rb, sb = ref.reduce_end_exact()
    assert np.array_equal(ra, rb) and sa == sb
    # estimator steps without an update: the accumulators
    eng.pgmc_steps(3, ids, 2)
    ref.pgmc_steps(3, ids, 2)
    assert np.array_equal(bits(eng.pg_get_accumulated(ids)), bits(ref.pg_get_accumulated(ids)))
    assert np.array_equal(bits(eng.get_parameters(ids[0])), bits(ref.get_parameters(ids[0])))
    eng.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
def test_two_learnable_moves_take_a_chain_of_launches(gpu, oracle, monkeypatch):
    """Several parameters AND several learnable moves on one shard (round 6): one launch per move, each with its own tail -- fold,
    gradients_data +=, the move's learning step --, the time step's sweep riding in the first: two launches where the records route
    took seven (sweep, two estimator launches, two accumulate and two update launches; 260 -> 202 us at 1e7 chains).  A different
    optimiser per move (the tail's record describes the CALL, the launch names its move), callbacks in the same stretch, steps without
    an update: the same bits as the records route and as the free-running oracle."""
    kinds, h0, h1 = [4, 2], [5e-3, 2e-2], [1e-6, 0.0]                       # NPG on move 1, BLPG on move 2
    eng, ref = _pair(gpu, oracle, 20011, DRIFT, 2, [[0.1, 0.3], [0.0, 0.8]], [0.5, 0.5])
    monkeypatch.setenv("AMC_NP_SMALL_LAUNCHES", "1")
    old, _ref2 = _pair(gpu, oracle, 20011, DRIFT, 2, [[0.1, 0.3], [0.0, 0.8]], [0.5, 0.5])
    monkeypatch.delenv("AMC_NP_SMALL_LAUNCHES")
    ids = [0, 1]
    assert eng.pg_route(2, 2, fused=True)[0] is False                       # one launch per learnable move, as the route says
    for stretch in (1, 2, 7):
        for e in (eng, old, ref):
            e.pgmc_steps(stretch, ids, 2, kinds, h0, h1)
        for k in ids:
            assert np.array_equal(bits(eng.get_parameters(k)), bits(ref.get_parameters(k))), (stretch, k)
            assert np.array_equal(bits(eng.get_parameters(k)), bits(old.get_parameters(k))), (stretch, k)
    for e in (eng, old, ref):
        e.pgmc_steps(3, ids, 1, kinds, h0, h1, reduce_begin=True)
    (ra, sa), (rb, sb), (rc, sc) = eng.reduce_end_exact(), ref.reduce_end_exact(), old.reduce_end_exact()
    assert np.array_equal(ra, rb) and np.array_equal(ra, rc) and sa == sb == sc
    for e in (eng, old, ref):
        e.pgmc_steps(2, ids, 3)                                              # estimator steps, no update: the accumulators fill
    acc = eng.pg_get_accumulated(ids)
    assert np.array_equal(bits(acc), bits(ref.pg_get_accumulated(ids))) and np.array_equal(bits(acc), bits(old.pg_get_accumulated(ids)))
    assert acc[0, -1] == 2 * 3 * 20011
    x = eng.download_state()[0]
    assert np.array_equal(bits(x), bits(ref.download_state()[0])) and np.array_equal(bits(x), bits(old.download_state()[0]))
    a, t = eng.download_counters()
    ao, to = ref.download_counters()
    assert np.array_equal(a, ao) and np.array_equal(t, to)
    assert eng.estimator_step == ref.estimator_step == old.estimator_step
    eng.close()
    old.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
def test_single_launch_time_step_of_a_three_parameter_policy(gpu, oracle):
    eng, ref = _pair(gpu, oracle, 9001, LEAN, 3, [[0.0, 0.6, -0.1]], [1.0])
    for stretch in (1, 7):
        eng.pgmc_steps(stretch, [0], 1, [4], [5e-3], [1e-6])          # NPG: the 3 x 3 metric inverted in the launch's tail
        ref.pgmc_steps(stretch, [0], 1, [4], [5e-3], [1e-6])
        assert np.array_equal(bits(eng.get_parameters(0)), bits(ref.get_parameters(0)))
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    eng.close()
    oracle.install_vector_policy(1, None)


@pytest.mark.gpu
def test_queued_parameter_read_returns_every_parameter(gpu):
    """amc_parameters_begin / amc_parameters_end_all: the read queued in stream order carries all P parameters of every move."""
    eng = gpu.HipEngine(n_chains=1000, device=0, potential="harmonic", beta=BETA, sigma=[[0.1, 0.3], [0.0, 0.8]], weight=[0.5, 0.5],
                        seed=2, proposal=DRIFT, n_params=2)
    eng.init_uniform(-1, 1)
    eng.parameters_begin()
    eng.pgmc_steps(5, [1], 1, [1], [1e-2], [0.0])            # learning steps queued BEHIND the read
    before = eng.parameters_end()
    assert before.shape == (2, 2) and np.array_equal(before, [[0.1, 0.3], [0.0, 0.8]])
    eng.parameters_begin()
    after = eng.parameters_end()
    assert np.array_equal(after[0], [0.1, 0.3]) and not np.array_equal(after[1], [0.0, 0.8])
    assert np.array_equal(after[1], eng.get_parameters(1))
    eng.close()
