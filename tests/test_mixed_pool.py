"""Pools that MIX policy and action types (amc_create_mixed_model; HipEngine(classes=..., class_of_move=...)).

In the reference every Move carries its own `action` and `policy` (src/metropolis.jl:140-162) and sample_action! /
log_proposal_density / perform_action! / invert_action! dispatch on their types; only across CHAINS must the moves of a pool
agree (:249-260).  The engine takes up to four expression sets ("classes") and a class per move.  The example pool:

  * move 1: the particle_1d Gaussian displacement (class GAUSS, written out as expressions),
  * move 2: a Langevin (drifted) proposal, whose mean leans on the state (class MALA),
  * move 3: a Gaussian displacement again, another sigma (class GAUSS),
  * move 4: a scaling action x -> x exp(delta) with the Jacobian in its log q (class SCALING: its own perform / invert).

The oracle evaluates the same expressions compiled by gcc, dispatching on the move's class (amo_set_policy_classes)."""
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
CLASSES = [GAUSS, MALA, SCALING]
CLASS_OF_MOVE = [0, 1, 0, 2]
SIGMA, WEIGHT = [0.3, 0.6, 0.1, 0.5], [0.3, 0.3, 0.2, 0.2]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _kw(M, **more):
    return dict(n_chains=M, potential="harmonic", beta=BETA, sigma=SIGMA, weight=WEIGHT, seed=21, classes=CLASSES,
                class_of_move=CLASS_OF_MOVE, **more)


def test_oracle_keeps_the_target_distribution_with_a_mixed_pool(oracle):
    """Started on x > 0 the scaling move never changes the sign, the displacements do: the pool samples the full Gaussian."""
    s = oracle.OracleSim(4000, **{k: v for k, v in _kw(4000).items() if k != "n_chains"})
    s.init_uniform(-2, 2)
    n, sx, sxx, _ = s.run_pooled_moments(3000, 400, 10, threads=8)
    assert sx / n == pytest.approx(0.0, abs=8e-3) and sxx / n == pytest.approx(1 / (2 * BETA), abs=4e-3)
    acc = s.acceptance()
    assert len(set(np.round(acc, 3))) == 4 and acc[2] > acc[0]       # four moves, four acceptance rates; the narrower Gaussian is accepted more
    oracle.install_policy_classes(None, None)


def test_argument_validation_needs_no_gpu(amc):
    kw = dict(n_chains=10, potential="harmonic", beta=BETA, sigma=[0.3, 0.6], weight=[0.5, 0.5])
    with pytest.raises(amc.AmcError, match="one class per move"):
        amc.HipEngine(classes=[GAUSS, MALA], class_of_move=[0], **kw)
    with pytest.raises(amc.AmcError, match="is no class"):
        amc.HipEngine(classes=[GAUSS, MALA], class_of_move=[0, 2], **kw)
    # (a class without a derivative expression is no error any more: the engine differentiates its logq, tests/test_autodiff.py)
    with pytest.raises(amc.AmcError, match="come together"):
        amc.HipEngine(classes=[GAUSS, SCALING[:4]], class_of_move=[0, 1], **kw)
    with pytest.raises(amc.AmcError, match=r"n_classes must be in \[1, 4\]"):
        amc.HipEngine(classes=[GAUSS] * 5, class_of_move=[0, 1], **kw)
    with pytest.raises(amc.AmcError, match="cannot be combined"):
        amc.HipEngine(classes=[GAUSS, MALA], class_of_move=[0, 1], proposal=MALA, **kw)


def _mixed_simulation(ma, engine_factory, path, device_resident=None, steps=30):
    chains = ma.ParticleChains.uniform(2001, BETA, 0.1, 2.0)
    pool = [ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), [0.3], 0.3),
            ma.Move(ma.Displacement(0.0), ma.ScriptPolicy(*MALA), [0.6], 0.3),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), [0.1], 0.2),
            ma.Move(ma.ScriptAction(perform=SCALING[3], invert=SCALING[4]), ma.ScriptPolicy(*SCALING[:3]), [0.5], 0.2)]
    opts = [ma.Static(), ma.VPG(0.05), ma.BLPG(0.02), ma.Static()]
    algos = [dict(algorithm=ma.Metropolis, pool=pool, seed=21, engine_factory=engine_factory),
             dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=opts, q_batch_size=2,
                  device_resident=device_resident),
             dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,), scheduler=list(range(0, steps + 1, 2)))]
    sim = ma.Simulation(chains, algos, steps, path=str(path))
    ma.run(sim)
    return chains, pool, sim


def test_canonical_gaussian_expressions_are_the_same_text_everywhere():
    """The sweep's table-row form of a Gaussian class (amc_model.h GaussRow) and the estimator's shortcut are switched on by TEXT:
    the class's expressions must be, character for character, the ones the host mirror writes the built-in policy out as.  The
    places that hold that text -- montecarlo_amd/metropolis.py, amc_rtc.hip, julia/AriannaHIP.jl, this file's GAUSS -- must agree."""
    import os
    from montecarlo_amd import metropolis as mp
    assert (mp.GAUSS_SAMPLE, mp.GAUSS_LOGQ, mp.GAUSS_DLOGQ) == GAUSS
    rtc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "montecarlo_amd", "csrc", "amc_rtc.hip")).read()
    jl = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "julia", "src", "AriannaHIP.jl")).read()
    for text in GAUSS:
        assert '"' + text + '"' in rtc, text
        assert '"' + text + '"' in jl, text            # AriannaHIP.GAUSS_CLASS


def test_host_mirror_builds_the_classes_of_a_mixed_pool(oracle, tmp_path):
    """Metropolis used to refuse a pool whose moves do not share one policy; it now hands the engine one class per distinct
    (policy, action) pair -- the built-in Gaussian written out as expressions -- and the run equals the oracle driven by hand."""
    import montecarlo_amd as ma
    chains, pool, sim = _mixed_simulation(ma, oracle.OracleEngine, tmp_path / "a", device_resident=True)
    eng = sim.algorithms[0].engine
    o = oracle.OracleEngine(**_kw(2001))
    o.init_uniform(0.1, 2.0)
    kinds, h0 = [1, 2], [0.05, 0.02]
    for t in range(1, 31):
        o.sweep(1)
        o.pg_accumulate([1, 2], 2)
        if t % 2 == 0:
            o.pg_update([1, 2], kinds, h0, [0.0, 0.0])
    assert np.array_equal(bits(chains.x), bits(o.download_state()[0]))
    assert [m.sigma for m in pool] == [o.get_parameters(k)[0] for k in range(4)]
    assert pool[0].sigma == 0.3 and pool[3].sigma == 0.5 and pool[1].sigma != 0.6 and pool[2].sigma != 0.1
    with pytest.raises(ValueError, match="at most 4 different"):
        ma.Metropolis(ma.ParticleChains.uniform(8, BETA), engine_factory=oracle.OracleEngine,
                      pool=[ma.Move(ma.Displacement(0.0), ma.ScaledGaussian(f"1.0 + {i}.0*x*x"), [0.3], 0.2) for i in range(5)])
    oracle.install_policy_classes(None, None)


# ---- GPU ----------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_sweeps_and_callback_sums_are_exact(gpu, oracle):
    eng, ref = gpu.HipEngine(device=0, **_kw(4099)), oracle.OracleEngine(**_kw(4099))
    for e in (eng, ref):
        e.init_uniform(-2.0, 2.0)
    for n in (1, 9, 50):
        eng.sweep(n)
        ref.sweep(n)
        assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0])), n
    a, t = eng.download_counters()
    ao, to = ref.download_counters()
    assert np.array_equal(a, ao) and np.array_equal(t, to)
    rec, steps = eng.reduce_exact()
    rec_o, steps_o = ref.reduce_exact()
    assert steps == steps_o and np.array_equal(rec, rec_o, equal_nan=True)
    eng.close()
    oracle.install_policy_classes(None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("M,q_batch", [(1, 1), (4099, 3), (50001, 1)])
def test_gradient_data_records_equal_the_oracles(gpu, oracle, M, q_batch):
    eng, ref = gpu.HipEngine(device=0, **_kw(M)), oracle.OracleEngine(**_kw(M))
    for e in (eng, ref):
        e.init_uniform(0.1, 2.0)
        e.sweep(3)
    for ids in ([1], [0, 1, 2, 3]):
        got, want = eng.pg_estimate_exact(ids, q_batch), ref.pg_estimate_exact(ids, q_batch)
        assert np.array_equal(got, want, equal_nan=True), ids
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    eng.close()
    oracle.install_policy_classes(None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("knob,pool", [("AMC_NO_GAUSS_CLASS_ROWS", "classes"), ("AMC_NO_SIGMA_MEMO", "classes"), ("AMC_NO_SIGMA_MEMO", "langevin_k2")])
def test_table_rows_and_the_sigma_memo_equal_the_expressions(gpu, monkeypatch, knob, pool):
    """Round 5, K > 1 sweeps of script-defined pools.  (a) A class whose expressions are the built-in Gaussian displacement's takes
    den = 2 sigma^2 and log(2 pi sigma^2)/2 from the move's table row instead of forming them per lane and step (amc_model.h
    GaussRow); (b) `amc_log(sigma)` in any expression reads log(sigma_k), formed once per launch and move, instead of fifty vector
    instructions per lane and step (SigmaArg).  The same operations on the same operands, so nothing may change: 3e5 chains, single-
    and multi-step launches, fused time steps, learning steps that move sigma (rows and memo must follow) -- with the shortcut and
    with its knob set."""
    if pool == "classes":
        kw, learn, kinds, h0 = _kw(300001), [0, 1, 2], [1, 1, 2], [0.05, 0.05, 0.02]
    else:
        kw = dict(n_chains=300001, potential="double_well", beta=BETA, sigma=[0.2, 0.6], weight=[0.3, 0.7], seed=5, proposal=MALA)
        learn, kinds, h0 = [0, 1], [1, 2], [0.02, 0.02]
    runs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv(knob, "1")
        else:
            monkeypatch.delenv(knob, raising=False)
        e = gpu.HipEngine(device=0, **kw)
        e.init_uniform(-2.0, 2.0)
        for _ in range(5):
            e.sweep(1)
        e.sweep(40)
        e.pgmc_steps(25, learn, 1, kinds, h0, [0.0] * len(learn))
        e.sweep(20)
        acc, tot = e.download_counters()
        runs.append((e.download_state()[0], acc, tot, [e.get_parameters(k)[0] for k in range(len(kw["sigma"]))]))
        e.close()
    assert runs[0][3] == runs[1][3] and all(runs[0][3][k] != kw["sigma"][k] for k in learn)        # the moves learned
    assert np.array_equal(bits(runs[0][0]), bits(runs[1][0]))
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])


@pytest.mark.gpu
def test_free_running_pgmc_and_shard_split(gpu, oracle):
    eng, ref = gpu.HipEngine(device=0, **_kw(20011)), oracle.OracleEngine(**_kw(20011))
    three = gpu.SplitEngine(device=0, n_parts=3, **_kw(20011))
    for e in (eng, ref, three):
        e.init_uniform(0.1, 2.0)
    ids, kinds, h0, h1 = [1, 2, 3], [1, 2, 4], [0.05, 0.02, 1e-3], [0.0, 0.0, 1e-6]
    for stretch in (1, 2, 10):
        eng.pgmc_steps(stretch, ids, 2, kinds, h0, h1)
        ref.pgmc_steps(stretch, ids, 2, kinds, h0, h1)
        for k in ids:
            assert eng.get_parameters(k)[0] == ref.get_parameters(k)[0], (stretch, k)
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    one = gpu.HipEngine(device=0, **_kw(20011))
    one.init_uniform(0.1, 2.0)
    for e in (one, three):
        e.sweep(4)
    assert np.array_equal(one.pg_estimate_exact(ids, 2), three.pg_estimate_exact(ids, 2))
    for e in (eng, one, three):
        e.close()
    oracle.install_policy_classes(None, None)


@pytest.mark.gpu
def test_host_mirror_on_the_engine_equals_the_test_double(gpu, oracle, tmp_path):
    import montecarlo_amd as ma
    ch_g, pool_g, _ = _mixed_simulation(ma, None, tmp_path / "gpu", device_resident=True)
    ch_o, pool_o, _ = _mixed_simulation(ma, oracle.OracleEngine, tmp_path / "cpu", device_resident=True)
    assert [m.sigma for m in pool_g] == [m.sigma for m in pool_o]
    assert np.array_equal(bits(ch_g.x), bits(ch_o.x))
    oracle.install_policy_classes(None, None)
