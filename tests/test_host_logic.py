"""Host-side mirror of the reference's driver (Simulation / run / schedules / callbacks / optimisers),
exercised on CPU through the engine_factory test seam with the oracle as the engine."""
import math
import os

import numpy as np
import pytest

import montecarlo_amd as ma
from montecarlo_amd import policy_guided as pg


def make_sim(oracle, path, M=10, steps=300, burn=50, pool=None, extra=(), fuse_sched=None, seed=42, sweepstep=1):
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = pool or (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 1.0),)
    sampletimes = ma.build_schedule(steps, burn, [0, 10])
    algorithm_list = (
        dict(algorithm=ma.Metropolis, pool=pool, seed=seed, sweepstep=sweepstep, engine_factory=oracle.OracleEngine),
        *extra,
        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
             scheduler=fuse_sched or sampletimes),
    )
    return ma.Simulation(chains, algorithm_list, steps, path=str(path)), pool


def test_build_schedule_matches_the_c_restatement(oracle):
    for args in [(10 ** 5, 1000, [0, 10]), (1000, 0, [0, 1, 5]), (100, 10, 30), (100, 10, 45), (10 ** 4, 100, 1000),
                 (1000, 10, 2.0), (10 ** 5, 1000, 10.0), (977, 13, [0, 3, 7])]:
        assert ma.build_schedule(*args) == oracle.build_schedule(*args), args
    with pytest.raises(ValueError):
        ma.build_schedule(1000, 10, 1.5)


def test_julia_repr():
    assert ma.julia_repr(0.25) == "0.25" and ma.julia_repr(1e-5) == "1.0e-5" and ma.julia_repr(1e6) == "1.0e6"
    assert ma.julia_repr(100000.0) == "100000.0" and ma.julia_repr(float("nan")) == "NaN"
    assert ma.julia_repr(np.array([0.5, float("nan")])) == "[0.5, NaN]" and ma.julia_repr(-0.0) == "-0.0"
    assert ma.julia_repr(0.1 + 0.2) == "0.30000000000000004" and ma.julia_repr(1234567.0) == "1.234567e6"


def test_config1_plumbing_rows_and_first_row(oracle, tmp_path):
    """MC_harmonic_oscillator.jl at reduced length: callback files get one row at t = 0 (store_first,
    acceptance = 0/0 = NaN) plus one per schedule entry; "$t $(value)" format."""
    sim, pool = make_sim(oracle, tmp_path, steps=2000, burn=100)
    ma.run(sim)
    energy = open(tmp_path / "energy.dat").read().splitlines()
    acc = open(tmp_path / "acceptance.dat").read().splitlines()
    n_sched = len(ma.build_schedule(2000, 100, [0, 10]))
    assert len(energy) == len(acc) == n_sched + 1
    assert acc[0] == "0 [NaN]" and energy[0].startswith("0 ")
    assert [int(r.split()[0]) for r in energy[1:4]] == [100, 110, 120] and energy[-1].startswith("2000 ")
    assert pool[0].total_calls == 10 * 2000 and 0 < pool[0].accepted_calls < pool[0].total_calls
    summary = open(tmp_path / "summary.log").read()
    assert "Metropolis" in summary and "Calls: 2000" in summary and "Simulation time" in summary
    assert sim.chains.x.shape == (10,) and np.array_equal(sim.chains.e, sim.chains.x ** 2)


def test_fused_run_equals_stepwise_run(oracle, tmp_path):
    """run(fuse=True) batches sweeps nobody observes; every row and the final state are identical."""
    rows = []
    for i, fuse in enumerate((False, True)):
        sim, _ = make_sim(oracle, tmp_path / str(i), steps=500, burn=40)
        ma.run(sim, fuse=fuse)
        rows.append((open(tmp_path / str(i) / "energy.dat").read(), open(tmp_path / str(i) / "acceptance.dat").read(),
                     sim.chains.x.copy()))
    assert rows[0][0] == rows[1][0] and rows[0][1] == rows[1][1] and np.array_equal(rows[0][2], rows[1][2])


def test_fusion_issues_few_launches(oracle, tmp_path):
    calls = []

    class Counting(oracle.OracleEngine):
        def sweep(self, n=1):
            calls.append(n)
            super().sweep(n)

    chains = ma.ParticleChains.uniform(4, 2.0)
    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 1.0),)
    al = (dict(algorithm=ma.Metropolis, pool=pool, engine_factory=Counting),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy,), scheduler=[100, 200, 300]))
    ma.run(ma.Simulation(chains, al, 300, path=str(tmp_path)))
    assert calls == [99, 1, 99, 1, 99, 1] and sum(calls) == 300


def test_run_asks_the_engine_for_the_sums_its_callbacks_read(oracle, tmp_path):
    """run() collects what the StoreCallbacks' callbacks read of a reduction (`needs`) and the sampler forms those sums only:
    the reference's own two callbacks need sum e alone; a callback that does not say what it reads keeps everything."""
    seen = []

    class Watching(oracle.OracleEngine):
        def set_reduce_columns(self, columns):
            seen.append(columns)
            super().set_reduce_columns(columns)

    def user_callback(simulation):
        return float(ma.callback_moments(simulation)[1])

    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 1.0),)
    cases = [((ma.callback_energy, ma.callback_acceptance), 1), ((ma.callback_acceptance,), 0), ((ma.callback_moments,), 6),
             ((ma.callback_energy, ma.callback_moments), 7), ((ma.callback_energy, user_callback), 7)]
    for i, (cbs, want) in enumerate(cases):
        seen.clear()
        al = (dict(algorithm=ma.Metropolis, pool=pool, seed=3, engine_factory=Watching),
              dict(algorithm=ma.StoreCallbacks, callbacks=cbs, scheduler=[0, 10]))
        sim = ma.Simulation(ma.ParticleChains.uniform(6, 2.0), al, 30, path=str(tmp_path / str(i)))
        ma.run(sim)
        assert seen == [want, 7]                      # ... and everything again once the run is over
        rows = sim.algorithms[1].rows
        assert all(np.all(np.isfinite(np.atleast_1d(v))) for r in rows for t, v in r if t > 0)     # what was asked for is there
    # without StoreCallbacks nothing is narrowed: a caller of Metropolis.reductions() gets every entry
    seen.clear()
    sim = ma.Simulation(ma.ParticleChains.uniform(6, 2.0), (dict(algorithm=ma.Metropolis, pool=pool, engine_factory=Watching),), 3,
                        path=str(tmp_path / "bare"))
    ma.run(sim)
    assert seen == [7, 7]


def test_callbacks_after_a_narrowed_run_give_real_values(oracle, tmp_path):
    """A run narrowed to sum e whose last step is a scheduled callback leaves a reduction of that state in the sampler's cache,
    formed without sum x and sum x^2.  finalise() widens the engine's columns again, and a callback evaluated after run!
    (in the reference: any f(simulation), src/algorithms.jl:97-102) must see real values, not the cached NaN entries."""
    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 1.0),)
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=3, engine_factory=oracle.OracleEngine),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy,), scheduler=[10, 20, 30]))
    sim = ma.Simulation(ma.ParticleChains.uniform(6, 2.0), al, 30, path=str(tmp_path))
    ma.run(sim)
    met = sim.algorithms[0]
    mom = ma.callback_moments(sim)
    assert np.all(np.isfinite(mom)), mom
    red = met.reductions()
    assert np.isfinite(red["mean_x"]) and np.isfinite(red["mean_x2"]) and np.isfinite(red["energy"])
    assert mom[1] == red["mean_x2"] and abs(red["energy"] - red["mean_x2"]) < 1e-12      # harmonic: e = x^2
    # ... and narrowing by hand between two reads of one state does not hand back the wider read's sibling with holes either
    met.set_reduction_needs(("energy",))
    assert np.isfinite(met.reductions()["energy"])
    met.set_reduction_needs(None)
    assert np.all(np.isfinite(ma.callback_moments(sim)))


def test_a_user_algorithm_in_the_list_keeps_every_sum(oracle, tmp_path):
    """Only algorithms this package defines are known not to read a reduction on the quiet: with a user's algorithm in the list
    nothing is narrowed, so callback_moments(simulation) inside its make_step is a number."""
    seen, got = [], []

    class Watching(oracle.OracleEngine):
        def set_reduce_columns(self, columns):
            seen.append(columns)
            super().set_reduce_columns(columns)

    class UserAlgorithm(ma.AriannaAlgorithm):
        def __init__(self, chains, **kw):
            pass

        def make_step(self, simulation):
            got.append(ma.callback_moments(simulation))

    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 1.0),)
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=3, engine_factory=Watching),
          dict(algorithm=UserAlgorithm, scheduler=[0, 10]),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy,), scheduler=[0, 10]))
    ma.run(ma.Simulation(ma.ParticleChains.uniform(6, 2.0), al, 30, path=str(tmp_path)))
    assert seen == [7, 7]
    assert got and all(np.all(np.isfinite(m)) for m in got)


def test_sweepstep_is_mc_steps_per_sweep(oracle, tmp_path):
    """metropolis.jl:205: one make_step! = sweepstep mc_step!s; sweepstep=3 x 100 sweeps == 300 single steps."""
    a, _ = make_sim(oracle, tmp_path / "a", steps=100, burn=10, sweepstep=3)
    b, _ = make_sim(oracle, tmp_path / "b", steps=300, burn=10, sweepstep=1)
    ma.run(a)
    ma.run(b)
    assert np.array_equal(a.chains.x, b.chains.x)


def test_dependencies_resolved_by_type(oracle, tmp_path):
    """simulation.jl:77-81: dependencies=(Metropolis,) becomes the previously built instance."""
    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.6),
            ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 0.4))
    extra = (dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.001))),
             dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
             dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=[50, 100]))
    sim, pool = make_sim(oracle, tmp_path, steps=100, burn=10, pool=pool, extra=extra)
    met, est, upd, prm, _cb = sim.algorithms
    assert est.metropolis is met and upd.estimator is est and est.learn_ids == [1]
    assert prm.parameters_list[1] is pool[1].parameters          # shared object, like the aliasing at metropolis.jl:252-260
    ma.run(sim)
    assert pool[0].sigma == 0.2                                  # Static never moves (estimator.jl:72)
    assert pool[1].sigma != 0.1 and met.engine.get_parameters(1)[0] == pool[1].sigma
    rows = open(tmp_path / "parameters" / "2" / "parameters.dat").read().splitlines()
    assert [r.split()[0] for r in rows] == ["0", "50", "100"] and rows[0] == "0 [0.1]"


def test_callbacks_require_exactly_one_metropolis(oracle, tmp_path):
    chains = ma.ParticleChains.uniform(4, 2.0)
    al = (dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy,)),)
    with pytest.raises(ValueError):
        ma.run(ma.Simulation(chains, al, 3, path=str(tmp_path)))


def test_constructor_errors(oracle):
    chains = ma.ParticleChains.uniform(4, 2.0)
    with pytest.raises(ValueError):
        ma.Metropolis(chains, pool=None, engine_factory=oracle.OracleEngine)
    with pytest.raises(TypeError):
        ma.Metropolis(chains, pool=("move",), engine_factory=oracle.OracleEngine)
    with pytest.raises(ValueError):
        ma.ParticleChains(4, 2.0, potential="lennard_jones")
    with pytest.raises(ValueError):
        ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1, 0.2], 1.0)
    with pytest.raises(AssertionError):
        ma.Simulation(chains, (dict(algorithm=ma.StoreCallbacks, callbacks=(), scheduler=[5, 3]),), 10, path="/tmp/x_amc")


# ---- optimisers: learning.jl vs the oracle's C restatement (P = 1) ----------------------------------------
@pytest.mark.parametrize("name,opt,h0,h1", [
    ("VPG", pg.VPG(1e-3), 1e-3, 0.0), ("BLPG", pg.BLPG(1e-3), 1e-3, 0.0), ("BLAPG", pg.BLAPG(1e-6, 1e-6), 1e-6, 1e-6),
    ("NPG", pg.NPG(1e-2, 1e-6), 1e-2, 1e-6), ("ANPG", pg.ANPG(1e-6, 1e-6), 1e-6, 1e-6),
    ("BLANPG", pg.BLANPG(1e-6, 1e-6), 1e-6, 1e-6), ("Static", pg.Static(), 0.0, 0.0)])
def test_learning_step_matches_oracle(oracle, name, opt, h0, h1):
    rng = np.random.default_rng(8)
    for _ in range(200):
        theta = rng.uniform(0.05, 2.0)
        j, dj, dlq, g = rng.uniform(0, 0.3), rng.normal(0, 0.2), rng.normal(0, 1), rng.uniform(0.1, 30)
        p = np.array([theta])
        gd = pg.GradientData(j, np.array([dj]), np.array([dlq]), np.array([[g]]), 1)
        pg.learning_step(p, gd, opt)
        want = oracle.learning_step(name, h0, h1, theta, [j, dj, dlq, g])
        assert p[0] == pytest.approx(want, rel=1e-14, abs=1e-300)


def test_gradient_data_algebra():
    a = pg.GradientData(1.0, np.array([2.0]), np.array([3.0]), np.array([[4.0]]), 2)
    s = a + a
    assert (s.j, s.grad_j[0], s.grad_logq_forward[0], s.g[0, 0], s.n) == (2.0, 4.0, 6.0, 8.0, 4)
    m = pg.average(s)
    assert (m.j, m.grad_j[0], m.g[0, 0], m.n) == (0.5, 1.0, 2.0, 4)
    z = pg.initialise_gradient_data(np.array([0.2]))
    assert z.j == 0.0 and z.n == 0 and z.g.shape == (1, 1)


def test_shard_range_even_boundaries_and_cover():
    for n in (1, 2, 7, 10, 101, 10 ** 7, 8 * 10 ** 7 + 3):
        for w in (1, 2, 3, 8):
            if n < 2 * w - 1:
                continue
            rs = [ma.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert all(a % 2 == 0 for a, _ in rs)
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 3      # one pair of imbalance + the odd tail chain


# ---- device-side stand-ins for the per-chain text I/O (storage.py) ---------------------------------------------
def test_histogram_snapshots_and_exact_resume(oracle, tmp_path):
    pool = lambda: (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.6),
                    ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 0.4))
    def build(path, steps, p):
        chains = ma.ParticleChains.uniform(300, 2.0)
        al = (dict(algorithm=ma.Metropolis, pool=p, seed=11, engine_factory=oracle.OracleEngine),
              dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.05))),
              dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
              dict(algorithm=ma.StoreHistogram, dependencies=(ma.Metropolis,), lo=-1.5, hi=1.5, bins=30, scheduler=list(range(10, steps + 1, 10))),
              dict(algorithm=ma.StoreSnapshots, dependencies=(ma.Metropolis,), stride=64, scheduler=list(range(20, steps + 1, 20))))
        return ma.Simulation(chains, al, steps, path=str(path))
    # one run of 120 steps ...
    p_full = pool()
    full = build(tmp_path / "full", 120, p_full)
    ma.run(full)
    hist = full.algorithms[3]
    assert int(hist.global_counts.sum()) == 300 * 12 and int(hist.global_counts[30 + 2]) == 0
    rows = open(tmp_path / "full" / "histogram.dat").read().splitlines()
    assert len(rows) == 31 and rows[0].startswith("# samples 3600")
    snaps = np.load(tmp_path / "full" / "snapshots_rank0.npy")
    ids = np.load(tmp_path / "full" / "snapshots_rank0_chain_ids.npy")
    assert snaps.shape == (7, 1 + 5) and list(snaps[:, 0]) == [0, 20, 40, 60, 80, 100, 120] and list(ids) == [0, 64, 128, 192, 256]
    assert np.array_equal(snaps[-1, 1:], full.chains.x[ids])
    # ... equals 50 steps + checkpoint + restore into a fresh object + 70 steps, bit for bit
    p_a = pool()
    a = build(tmp_path / "a", 50, p_a)
    ma.run(a)
    ma.checkpoint(a.algorithms[0], str(tmp_path / "ckpt"), estimator=a.algorithms[1])
    p_b = pool()
    b = build(tmp_path / "b", 70, p_b)
    ma.restore(b.algorithms[0], str(tmp_path / "ckpt"), estimator=b.algorithms[1])
    ma.run(b)
    assert np.array_equal(b.chains.x, full.chains.x)
    assert [m.sigma for m in p_b] == [m.sigma for m in p_full] and p_b[1].sigma != 0.1
    assert [(m.accepted_calls, m.total_calls) for m in p_b] == [(m.accepted_calls, m.total_calls) for m in p_full]
    with pytest.raises(ValueError):
        other = build(tmp_path / "c", 10, pool())
        other.algorithms[0].seed = 12
        ma.restore(other.algorithms[0], str(tmp_path / "ckpt"))


def test_two_histograms_on_one_metropolis(oracle, tmp_path):
    """The engine keeps one running histogram: the first StoreHistogram to sample owns it, a second one -- here with the SAME
    bins, which used to add into the same accumulator and double the first one's counts -- fetches its counts each time."""
    chains = ma.ParticleChains.uniform(500, 2.0)
    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.3], 1.0),)
    sched = list(range(5, 41, 5))
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=5, engine_factory=oracle.OracleEngine),
          dict(algorithm=ma.StoreHistogram, dependencies=(ma.Metropolis,), lo=-1.5, hi=1.5, bins=20, scheduler=sched, path=str(tmp_path / "h1")),
          dict(algorithm=ma.StoreHistogram, dependencies=(ma.Metropolis,), lo=-1.5, hi=1.5, bins=20, scheduler=sched, path=str(tmp_path / "h2")),
          dict(algorithm=ma.StoreHistogram, dependencies=(ma.Metropolis,), lo=-1.0, hi=1.0, bins=8, scheduler=sched[1::2], path=str(tmp_path / "h3")))
    sim = ma.Simulation(chains, al, 40, path=str(tmp_path))
    ma.run(sim)
    h1, h2, h3 = sim.algorithms[1:4]
    assert int(h1.global_counts.sum()) == int(h2.global_counts.sum()) == 500 * len(sched)
    assert np.array_equal(h1.global_counts, h2.global_counts) and h1.mean == h2.mean
    assert int(h3.global_counts.sum()) == 500 * len(sched[1::2])


def test_resume_keeps_gradient_accumulators_when_updates_are_sparser_than_the_estimator(oracle, tmp_path):
    """PolicyGradientUpdate every 7 steps, estimator every step (update.jl:50-57 averages everything accumulated since the
    last update): a checkpoint that falls between two updates must carry the device-resident gradients_data, or the
    first update after the resume averages the post-resume samples only and sigma leaves the uninterrupted run."""
    pool = lambda: (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.6),
                    ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 0.4))
    def build(path, steps, p, updates):
        chains = ma.ParticleChains.uniform(200, 2.0)
        al = (dict(algorithm=ma.Metropolis, pool=p, seed=3, engine_factory=oracle.OracleEngine),
              dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.05)), q_batch_size=2),
              dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,), scheduler=updates))
        return ma.Simulation(chains, al, steps, path=str(path))
    p_full = pool()
    full = build(tmp_path / "full", 60, p_full, list(range(7, 57, 7)) + [60])     # updates at t = 7, 14, ..., 56, 60
    ma.run(full)
    p_a = pool()
    a = build(tmp_path / "a", 25, p_a, [7, 14, 21])          # the first 25 steps of the same run
    ma.run(a)
    est_a = a.algorithms[1]
    assert est_a.device_resident
    est_a.refresh()
    assert est_a.gradients_data[0].n == 4 * 200 * 2          # steps 22..25 are pending in the accumulator
    ma.checkpoint(a.algorithms[0], str(tmp_path / "ckpt"), estimator=est_a)
    p_b = pool()
    b = build(tmp_path / "b", 35, p_b, list(range(3, 32, 7)) + [35])              # t' = 3, ..., 31, 35 <-> t = 28, ..., 56, 60
    ma.restore(b.algorithms[0], str(tmp_path / "ckpt"), estimator=b.algorithms[1])
    ma.run(b)
    assert np.array_equal(b.chains.x, full.chains.x)
    assert [m.sigma for m in p_b] == [m.sigma for m in p_full] and p_b[1].sigma != 0.1


def _pgmc_sim(oracle, path, steps, update_sched=None, factory=None):
    chains = ma.ParticleChains.uniform(12, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.6),
            ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 0.4))
    upd = dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,))
    if update_sched is not None:
        upd["scheduler"] = update_sched
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=5, engine_factory=factory or oracle.OracleEngine),
          dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.01)),
               q_batch_size=2),
          upd,
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
               scheduler=ma.build_schedule(steps, 20, 25)))
    return ma.Simulation(chains, al, steps, path=str(path)), pool


@pytest.mark.parametrize("update_sched", [None, "every2"])
def test_grouped_pgmc_steps_equal_stepwise(oracle, tmp_path, update_sched):
    """run(fuse=True) issues [Metropolis, estimator, update] time steps as one engine call per run of such steps
    (PGMC_harmonic_oscillator.jl:24-33 schedules all three every t); rows, state and sigma are those of fuse=False."""
    steps = 120
    sched = None if update_sched is None else ma.build_schedule(steps, 10, 2)       # pgmc_test.jl:38: update every 2
    out = []
    for i, fuse in enumerate((False, True)):
        sim, pool = _pgmc_sim(oracle, tmp_path / str(i), steps, sched)
        ma.run(sim, fuse=fuse)
        out.append((open(tmp_path / str(i) / "energy.dat").read(), open(tmp_path / str(i) / "acceptance.dat").read(),
                    sim.chains.x.copy(), pool[1].sigma, getattr(sim.algorithms[0].engine, "pgmc_calls", 0)))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]
    assert out[0][3] != 0.1                          # the learnable sigma moved
    assert out[0][4] == 0 and out[1][4] > 0          # only the fused run used the grouped call


def test_grouped_pgmc_issues_few_engine_calls(oracle, tmp_path):
    calls = []

    class Counting(oracle.OracleEngine):
        def pgmc_steps(self, n, *a, **k):
            calls.append(n)
            super().pgmc_steps(n, *a, **k)

    sim, _ = _pgmc_sim(oracle, tmp_path, 100, factory=Counting)
    ma.run(sim)
    # callbacks at 20, 45, 70, 95, 100 cut the run into groups that END WITH the observed step: StoreCallbacks comes after
    # the three in the list, so it runs behind the group and sees the state it leaves.  A group is two engine calls: its
    # first n - 1 steps, then -- after the previous callback's sums have been fetched -- the step that forms the next ones.
    assert calls == [19, 1, 24, 1, 24, 1, 24, 1, 4, 1]


def test_deferred_parameter_rows_are_the_same_rows_written_one_period_later(oracle, tmp_path):
    """StoreParameters beside device-resident learning steps (the reference's PGMC script stores the parameters on the
    callbacks' schedule, PGMC_harmonic_oscillator.jl:30): defer=True queues the read at t (engine.parameters_begin) and writes
    the row when the next scheduled time comes or at finalise; files and rows are those of defer=False, and only one read is
    ever in flight (the test double asserts it)."""
    out = []
    for i, defer in enumerate((False, True)):
        steps, path = 100, str(tmp_path / str(i))
        chains = ma.ParticleChains.uniform(12, 2.0, -2.0, 2.0)
        pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.6), ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 0.4))
        reads = []

        class Watching(oracle.OracleEngine):
            def parameters_begin(self):
                reads.append("begin")
                super().parameters_begin()

            def parameters_end(self):
                reads.append("end")
                return super().parameters_end()

        al = (dict(algorithm=ma.Metropolis, pool=pool, seed=5, engine_factory=Watching),
              dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.01)), q_batch_size=2),
              dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
              dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy,), scheduler=ma.build_schedule(steps, 20, 25)),
              dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=ma.build_schedule(steps, 20, 25), defer=defer,
                   store_last=True))
        sim = ma.Simulation(chains, al, steps, path=path)
        ma.run(sim)
        files = [open(os.path.join(path, "parameters", str(k + 1), "parameters.dat")).read() for k in range(2)]
        rows = [[(t, float(v[0])) for t, v in r] for r in sim.algorithms[-1].rows]
        out.append((files, rows, reads, pool[1].sigma))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert out[0][2] == [] and out[1][2] == ["begin", "end"] * 5          # times 20, 45, 70, 95, 100: one read in flight
    assert len(out[1][1][1]) == 7 and out[1][1][1][0] == (0, 0.1) and out[1][1][1][-1][1] == out[1][3] != 0.1   # t = 0 row, five, store_last
    assert out[0][3] == out[1][3]


def test_two_store_parameters_on_one_metropolis_share_the_queued_read(oracle, tmp_path):
    """Two StoreParameters on the same Metropolis (different ids, paths, schedules) beside device-resident learning steps: the
    engine keeps one parameter read in flight, so instances that are due at the same t share it (Metropolis.parameters_async)
    and one that is due at a later t fetches the older read first -- rows and files equal those of reading at once."""
    out = []
    for i, defer in enumerate((False, True)):
        steps, path = 100, str(tmp_path / str(i))
        chains = ma.ParticleChains.uniform(12, 2.0, -2.0, 2.0)
        pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.6), ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 0.4))
        al = (dict(algorithm=ma.Metropolis, pool=pool, seed=5, engine_factory=oracle.OracleEngine),
              dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.VPG(0.02), ma.VPG(0.01)), q_batch_size=2),
              dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
              dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), ids=[0], scheduler=ma.build_schedule(steps, 20, 20), defer=defer,
                   path=os.path.join(path, "a"), store_last=True),
              dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), ids=[1], scheduler=ma.build_schedule(steps, 20, 30), defer=defer,
                   path=os.path.join(path, "b")))
        sim = ma.Simulation(chains, al, steps, path=path)
        ma.run(sim)
        rows = [[[(t, float(v[0])) for t, v in r] for r in alg.rows] for alg in sim.algorithms[-2:]]
        out.append(rows)
    assert out[0] == out[1]
    assert [t for t, _ in out[1][0][0]] == [0, 20, 40, 60, 80, 100, 100] and [t for t, _ in out[1][1][0]] == [0, 20, 50, 80, 100]
    assert out[1][0][0][1][1] != 0.2 and out[1][1][0][1][1] != 0.1                 # both moves learned


def test_deferred_callback_rows_are_the_same_rows_written_one_period_later(oracle, tmp_path):
    """StoreCallbacks(defer=True, the default for the engine-backed callbacks): the row of time t is written when the next
    scheduled time comes (or at finalise); the reduction is claimed at t (a ticket) and fetched later.  Files and rows are
    those of defer=False; during the run the rows lag one entry; at most TWO reductions are in flight on the engine (the sums
    of one observation point are being formed while the host reads those of the one before), fetched oldest first."""
    in_flight = []

    class Watching(oracle.OracleEngine):
        def reduce_begin(self):              # sweep_reduce_begin ends here too
            assert len(getattr(self, "_pending", [])) < 2, "a third reduction was begun while two were in flight"
            super().reduce_begin()
            in_flight.append("begin")

        def reduce_end_exact(self):
            in_flight.append("end")
            return super().reduce_end_exact()

    out = []
    for i, defer in enumerate((False, True)):
        chains = ma.ParticleChains.uniform(8, 2.0)
        pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.2], 0.7), ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.6], 0.3))
        al = (dict(algorithm=ma.Metropolis, pool=pool, seed=9, engine_factory=Watching),
              dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance, ma.callback_moments),
                   scheduler=ma.build_schedule(90, 10, 10), defer=defer, store_last=True))
        sim = ma.Simulation(chains, al, 90, path=str(tmp_path / str(i)))
        cb = sim.algorithms[1]
        assert cb.defer is defer
        seen = []
        orig = cb.make_step

        def spy(simulation, orig=orig, cb=cb, seen=seen):
            orig(simulation)
            seen.append((simulation.t, len(cb.rows[0])))
        cb.make_step = spy
        del in_flight[:]
        ma.run(sim)
        files = [open(tmp_path / str(i) / f).read() for f in ("energy.dat", "acceptance.dat", "moments.dat")]
        out.append((files, [list(r) for r in cb.rows], seen, list(in_flight)))
    assert out[0][0] == out[1][0]                                     # byte-identical files
    assert [len(r) for r in out[1][1]] == [11, 11, 11]                # t = 0, 10 .. 90, store_last at 90
    # rows written by the time make_step(t) returns: at once = all up to t; deferred = one behind
    assert [n for _, n in out[0][2]] == list(range(1, 12)) and [n for _, n in out[1][2]] == list(range(0, 11))
    # deferred: nothing was thrown away unread, and the sums of time t were fetched only after those of the next scheduled
    # time had been begun -- the host never waits for the reduction it has just queued
    assert out[1][3].count("begin") == out[1][3].count("end") == 10   # t = 0, then the nine scheduled times (store_last re-reads t = 90's)
    assert out[1][3] == ["begin"] + ["begin", "end"] * 9 + ["end"]
    assert out[0][3] == ["begin", "end"] * 10                         # at once: begun and fetched on the spot
    # a callback without a deferred form keeps the whole list at-once
    sim = ma.Simulation(ma.ParticleChains.uniform(4, 2.0), (dict(algorithm=ma.Metropolis, pool=pool, engine_factory=oracle.OracleEngine),
                        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, lambda s: 1.0), scheduler=[5])), 5,
                        path=str(tmp_path / "mixed"))
    assert sim.algorithms[1].defer is False


def test_grouped_pgmc_steps_carry_the_callback_sums_of_their_last_step(oracle, tmp_path):
    """A callback scheduled at the last step of a [Metropolis, estimator, update] group observes the state the group leaves:
    run() asks the engine for the sums with the group (pgmc_steps(..., reduce_begin=True) = amc_pgmc_steps_reduce_begin), no
    separate reduction pass; rows equal the stepwise run's."""
    calls = []

    class Counting(oracle.OracleEngine):
        def pgmc_steps(self, n, *a, reduce_begin=False, **k):
            calls.append((n, reduce_begin))
            super().pgmc_steps(n, *a, reduce_begin=reduce_begin, **k)

        def reduce(self):
            calls.append("reduce")
            return super().reduce()

        def reduce_begin(self):
            calls.append("reduce_begin")
            super().reduce_begin()

    sim, pool = _pgmc_sim(oracle, tmp_path / "g", 100, factory=Counting)
    ma.run(sim)
    grouped = [c for c in calls if isinstance(c, tuple)]
    # each group: its first n - 1 steps, then the observed step with the sums (the previous sums are fetched in between)
    assert grouped == [(19, False), (1, True), (24, False), (1, True), (24, False), (1, True), (24, False), (1, True), (4, False), (1, True)]
    # the only reductions outside the groups: t = 0 (store_first) -- OracleEngine.pgmc_steps itself calls reduce_begin -> reduce
    assert calls.count("reduce_begin") == 5 + 1
    ref, _ = _pgmc_sim(oracle, tmp_path / "s", 100)
    ma.run(ref, fuse=False)
    assert open(tmp_path / "g" / "energy.dat").read() == open(tmp_path / "s" / "energy.dat").read()
    assert open(tmp_path / "g" / "acceptance.dat").read() == open(tmp_path / "s" / "acceptance.dat").read()


def test_bench_cpu_baseline_reports_the_cpus_it_was_granted(oracle, monkeypatch):
    """bench.py's cpu_baseline: `cores` is what the sample really had -- the fastest thread count of the ladder cut by the
    cgroup quota (BENCH_r05 said "cores: 64" for 64 threads on a 16-CPU quota); `threads` keeps the thread count, the ladder
    and the quota ride along, and the sample line says under which quota it ran."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "M_PER_GPU", 4000)              # a small ensemble: the 4000-sweep cap ends every timing leg in well under a second
    monkeypatch.setattr(bench, "cpu_allotment", lambda: (64, 3.0))
    line = bench.cpu_baseline(budget_s=0.2)
    assert line["kind"] == "port" and line["unit"] == "chain-updates/s" and line["value"] > 0 and line["single_thread_value"] > 0
    assert line["cpu_quota"] == 3.0 and line["threads"] >= 2 and str(line["threads"]) in line["thread_ladder"]
    assert line["cores"] == min(line["threads"], 3)
    assert "quota of 3 CPUs" in line["sample"] and f"on {line['threads']} threads" in line["sample"]
    monkeypatch.setattr(bench, "cpu_allotment", lambda: (64, None))
    line = bench.cpu_baseline(budget_s=0.2)
    assert line["cores"] == line["threads"] and "quota" not in line["sample"]
