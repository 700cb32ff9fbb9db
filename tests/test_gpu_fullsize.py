"""BASELINE.json's full sizes (M = 1e7 chains per GPU) on the HIP path: size-independent properties,
the reference's statistical known answers tightened by the ensemble size, and the host driver
(Simulation / run / callbacks) running end to end on the device against the oracle engine."""
import json
import math
import os

import numpy as np
import pytest

import montecarlo_amd as ma

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
M_FULL = 10_000_000


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_config2_harmonic_statistics_at_full_size(gpu):
    """BASELINE config 2: harmonic, beta = 2, sigma = 0.1, M = 1e7.  The reference's targets
    (distribution_test.jl:36-37, atol 1e-3) hold with margin at this ensemble size."""
    e = gpu.HipEngine(n_chains=M_FULL, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1,
                      per_chain_counters=False)
    e.init_uniform(-2, 2)
    e.sweep(3000)                                   # burn-in, fused
    acc0, tot0 = e.counter_totals()
    sx = sxx = se = 0.0
    n = 20
    for _ in range(n):
        e.sweep(100)
        r = e.reduce()
        se += r[0] / M_FULL
        sx += r[1] / M_FULL
        sxx += r[2] / M_FULL
    acc1, tot1 = e.counter_totals()
    mean, var = sx / n, sxx / n - (sx / n) ** 2
    assert mean == pytest.approx(0.0, abs=5e-4)
    assert math.sqrt(var) == pytest.approx(0.5, abs=5e-4)
    assert se / n == pytest.approx(0.25, abs=5e-4)                                     # <e> = 1/(2 beta)
    rate = (acc1[0] - acc0[0]) / (tot1[0] - tot0[0])
    assert rate == pytest.approx(KATS["analytic"]["acceptance"]["beta=2.0,sigma=0.1"], abs=3e-4)
    assert tot1[0] == M_FULL * 5000
    e.close()


def test_config3_double_well_mixed_pool_at_full_size(gpu):
    """BASELINE config 3: U = (x^2 - 1)^2, beta = 2, two sigmas (0.1, 1.0), weights (0.5, 0.5), M = 1e7.
    Known answers by quadrature: <U> = 0.272864, <x^2> = 0.852136."""
    from scipy import integrate
    z = integrate.quad(lambda x: math.exp(-2 * (x * x - 1) ** 2), -4, 4)[0]
    mean_u = integrate.quad(lambda x: (x * x - 1) ** 2 * math.exp(-2 * (x * x - 1) ** 2), -4, 4)[0] / z
    mean_x2 = integrate.quad(lambda x: x * x * math.exp(-2 * (x * x - 1) ** 2), -4, 4)[0] / z
    k = KATS["analytic"]["double_well_beta2"]
    assert mean_u == pytest.approx(k["mean_U"], abs=1e-6) and mean_x2 == pytest.approx(k["mean_x2"], abs=1e-6)
    e = gpu.HipEngine(n_chains=M_FULL, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
    e.init_uniform(-2, 2)
    e.sweep(400)
    r = e.reduce()
    assert r[0] / M_FULL == pytest.approx(mean_u, abs=1.5e-3)
    assert r[2] / M_FULL == pytest.approx(mean_x2, abs=1.5e-3)
    assert r[1] / M_FULL == pytest.approx(0.0, abs=1.5e-3)
    acc, tot = e.counter_totals()
    assert tot.sum() == M_FULL * 400 and abs(tot[0] / tot.sum() - 0.5) < 1e-3           # categorical pick is fair
    ratio = r[4:] / M_FULL
    assert np.allclose(ratio, acc / tot, atol=2e-3) and ratio[0] > ratio[1]               # small steps accept more
    e.close()


def test_full_size_properties(gpu):
    """Properties that need no oracle run at 1e7: determinism, fused == stepwise, shard invariance,
    e == potential(x), counter conservation."""
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=11)
    a = gpu.HipEngine(n_chains=M_FULL, **kw)
    b = gpu.HipEngine(n_chains=M_FULL, **kw)
    a.init_uniform(-2, 2)
    b.init_uniform(-2, 2)
    a.sweep(12)
    for _ in range(12):
        b.sweep(1)
    xa, ea = a.download_state()
    xb, _ = b.download_state()
    assert np.array_equal(bits(xa), bits(xb))                       # fused == stepwise, run-to-run deterministic
    assert np.array_equal(ea, xa * xa)
    acc_a, tot_a = a.download_counters()
    assert tot_a.min() == tot_a.max() == 12 and acc_a.max() <= 12
    assert acc_a.sum() == a.counter_totals()[0][0] == b.counter_totals()[0][0]
    b.close()
    # two half-size shards reproduce the whole
    half = M_FULL // 2
    parts = []
    for off in (0, half):
        p = gpu.HipEngine(n_chains=half, chain_offset=off, n_chains_global=M_FULL, **kw)
        p.init_uniform(-2, 2)
        p.sweep(12)
        parts.append(p.download_state()[0])
        p.close()
    assert np.array_equal(bits(np.concatenate(parts)), bits(xa))
    # reductions are deterministic (two-pass, fixed order) and consistent with a host sum
    r1, r2 = a.reduce(), a.reduce()
    assert np.array_equal(bits(r1), bits(r2))
    assert r1[0] == pytest.approx(float(np.sum(ea)), rel=1e-11) and r1[1] == pytest.approx(float(np.sum(xa)), abs=1e-6)
    assert r1[4] / M_FULL == pytest.approx(acc_a.sum() / (12 * M_FULL), rel=1e-12)
    a.close()


def test_slice_of_full_size_run_matches_oracle(gpu, oracle):
    """Chains are independent and keyed by global id: an oracle run of a 4096-chain slice taken from the
    middle of the 1e7 ensemble must equal the same slice of the device run, bit for bit."""
    kw = dict(potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=123)
    e = gpu.HipEngine(n_chains=M_FULL, **kw)
    e.init_uniform(-2, 2)
    e.sweep(20)
    x, _ = e.download_state()
    acc, tot = e.download_counters()
    for off in (0, 5_000_000, M_FULL - 4096):
        o = oracle.OracleSim(4096, chain_offset=off, **kw)
        o.init_uniform(-2, 2)
        o.make_steps(20, threads=4)
        assert np.array_equal(bits(x[off:off + 4096]), bits(o.state()[0]))
        ao, to = o.counters()
        assert np.array_equal(acc[:, off:off + 4096], ao) and np.array_equal(tot[:, off:off + 4096], to)
    e.close()


def test_ensemble_beyond_2_to_the_32_chains(gpu, oracle):
    """Maximum sizes: 2^32 + 2^20 + 1 chains on one device (34 GB of state out of 288) -- chain and pair indices pass
    2^31 and 2^32, the last chain is a lone one.  Slices at the start, across the 2^32 boundary and at the very end must
    equal the oracle's runs of the same global chains, bit for bit; the callback sums stay consistent."""
    M = (1 << 32) + (1 << 20) + 1
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.3], weight=[1.0], seed=77)
    e = gpu.HipEngine(n_chains=M, per_chain_counters=False, **kw)
    e.init_uniform(-2, 2)
    for n in (1, 1, 6):                      # single-step launches and a fused one
        e.sweep(n)
    for off, cnt in ((0, 1024), ((1 << 32) - 512, 1024), (M - 1025, 1025)):
        assert off % 2 == 0
        o = oracle.OracleSim(cnt, chain_offset=off, **kw)
        o.init_uniform(-2, 2)
        o.make_steps(8, threads=4)
        assert np.array_equal(bits(e.download_strided(off, 1, cnt)), bits(o.state()[0]))
    red = e.reduce()
    assert red[3] == float(M)
    assert red[0] / M == pytest.approx(red[2] / M, rel=1e-12)                   # harmonic: e = x^2
    assert 0.2 < red[0] / M < 1.4 and 0.5 < red[4] / M < 1.0                    # far from equilibrium after 8 sweeps, but sane
    acc, tot = e.counter_totals()
    assert int(tot[0]) == 8 * M and int(acc[0]) == int(round(red[4] * 8))
    e.close()


# ---- the host driver on the device -------------------------------------------------------------------------------
def _run_config1(engine_factory, path, steps=3000):
    chains = ma.ParticleChains.uniform(10, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 1.0),)
    sampletimes = ma.build_schedule(steps, 1000, [0, 10])
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42, parallel=False, engine_factory=engine_factory),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=sampletimes))
    sim = ma.Simulation(chains, al, steps, path=str(path))
    ma.run(sim)
    return sim, pool


def test_config1_driver_on_device_matches_oracle_engine(gpu, oracle, tmp_path):
    """BASELINE config 1 (MC_harmonic_oscillator.jl: M = 10, beta = 2, sigma = 0.1, seed 42) through
    Simulation / run / StoreCallbacks on the GPU; same rows as the oracle-backed run."""
    dev, pool_d = _run_config1(None, tmp_path / "hip")
    ref, pool_r = _run_config1(oracle.OracleEngine, tmp_path / "cpu")
    assert isinstance(dev.algorithms[0].engine, gpu.HipEngine)
    assert np.array_equal(bits(dev.chains.x), bits(ref.chains.x)) and np.array_equal(bits(dev.chains.e), bits(ref.chains.e))
    assert (pool_d[0].accepted_calls, pool_d[0].total_calls) == (pool_r[0].accepted_calls, pool_r[0].total_calls)
    rows_d, rows_r = dev.algorithms[1].rows, ref.algorithms[1].rows
    assert [t for t, _ in rows_d[0]] == [t for t, _ in rows_r[0]] and len(rows_d[0]) == 202
    np.testing.assert_allclose([v for _, v in rows_d[0]], [v for _, v in rows_r[0]], rtol=1e-13)
    np.testing.assert_allclose(np.array([v for _, v in rows_d[1]]), np.array([v for _, v in rows_r[1]]), rtol=1e-13,
                               equal_nan=True)
    assert open(tmp_path / "hip" / "acceptance.dat").readline() == "0 [NaN]\n"


def test_config5_pgmc_learns_sigma_on_device(gpu, tmp_path):
    """BASELINE config 5 at reduced M: PGMC_harmonic_oscillator.jl pool (sigma 0.2 / 0.1, weights 0.6 / 0.4),
    optimisers (Static, VPG); with 1e5 chains the gradient is ~1e4x less noisy than the example's M = 10,
    so a larger eta reaches the plateau sigma* ~ 1.2 (pgmc_test.jl:50, atol 0.2) in 400 updates."""
    M, steps = 100_000, 400
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, 0.6),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.4))
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42),
          dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.5))),
          dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
               scheduler=ma.build_schedule(steps, 200, 20)))
    sim = ma.Simulation(chains, al, steps, path=str(tmp_path))
    ma.run(sim)
    assert pool[0].sigma == 0.2                                             # Static stays exactly
    assert pool[1].sigma == pytest.approx(KATS["pgmc"]["sigma_star"], abs=KATS["pgmc"]["sigma_atol"])
    assert sim.algorithms[0].engine.get_parameters(1)[0] == pool[1].sigma   # device copy follows the host array
    energies = [v for t, v in sim.algorithms[3].rows[0] if t >= 200]
    assert np.mean(energies) == pytest.approx(0.25, abs=5e-3)


def _run_pgmc(device_resident, path, optimisers, weights, M=20_000, steps=60, q=3, engine_factory=None):
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = tuple(ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, w) for w in weights)
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42, engine_factory=engine_factory),
          dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=optimisers, q_batch_size=q,
               device_resident=device_resident),
          dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,), scheduler=ma.build_schedule(steps, 10, 2)),
          dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=ma.build_schedule(steps, 10, 10)))
    sim = ma.Simulation(chains, al, steps, path=str(path))
    ma.run(sim)
    return sim, pool


def test_device_resident_learning_matches_host_path_and_oracle(gpu, oracle, tmp_path):
    """amc_pg_accumulate / amc_pg_update (gradients_data and learning_step! kept on the GPU) against the host
    path (amc_pg_estimate + numpy learning_step!) and against the oracle-backed run, all 7 optimisers."""
    weights = KATS["pgmc"]["weights"]
    optimisers = (ma.Static(), ma.VPG(0.05), ma.BLPG(0.05), ma.BLAPG(1e-4, 1e-6), ma.NPG(1e-2, 1e-6), ma.ANPG(1e-4, 1e-6),
                  ma.BLANPG(1e-4, 1e-6))
    dev, pool_d = _run_pgmc(True, tmp_path / "dev", optimisers, weights)
    host, pool_h = _run_pgmc(False, tmp_path / "host", optimisers, weights)
    ref, pool_r = _run_pgmc(False, tmp_path / "ref", optimisers, weights, engine_factory=oracle.OracleEngine)
    assert dev.algorithms[1].device_resident and not host.algorithms[1].device_resident
    sd, sh, sr = ([m.sigma for m in p] for p in (pool_d, pool_h, pool_r))
    assert sd[0] == sh[0] == sr[0] == 0.2 and all(s != 0.2 for s in sd[1:])
    np.testing.assert_allclose(sd, sh, rtol=1e-12)
    np.testing.assert_allclose(sh, sr, rtol=1e-9)          # device tree sums vs the oracle's sequential fold
    np.testing.assert_allclose(dev.chains.x, host.chains.x, rtol=0, atol=1e-9)
    rows_d = open(tmp_path / "dev" / "parameters" / "2" / "parameters.dat").read().splitlines()
    rows_h = open(tmp_path / "host" / "parameters" / "2" / "parameters.dat").read().splitlines()
    assert [r.split()[0] for r in rows_d] == [r.split()[0] for r in rows_h] == ["0", "10", "20", "30", "40", "50", "60"]
    eng = dev.algorithms[0].engine
    assert np.all(eng.pg_get_accumulated([1, 2, 3, 4, 5, 6])[:, 4] == 0)        # reset after the last update
    with pytest.raises(gpu.AmcError, match="sigma outside"):
        eng.pg_accumulate([1], 1)
        eng.pg_update([1], [1], [-1e6], [0.0])                                   # VPG with a huge negative eta
        eng.pg_get_accumulated([1])
    assert eng.get_parameters(1)[0] == pool_d[1].sigma                           # the bad step was not applied


def test_config4_eight_shards_of_1e7_on_one_device(gpu, oracle):
    """BASELINE config 4 (particle_1d harmonic, 8 x 1e7 chains sharded over 8 GPUs, energy / acceptance callbacks
    all-reduced every 10 sweeps; the reference maps mc_sweep! over eachindex(chains), src/metropolis.jl:302-309) with the
    eight shards as eight handles on ONE device: chain_offset = r 1e7, n_chains_global = 8e7, the callback sums formed in
    the sweep launches and MERGED as exact integer records (amc_reduce_end_exact + amc_xsum_merge: what
    sharding.allreduce_xsum does across processes).  Checks: slices of every shard equal the oracle's
    run of the same GLOBAL chain ids bit for bit; the sharded ensemble equals ONE handle holding all 8e7 chains -- positions
    bit for bit, accepted totals exactly, and every callback sum `==`: the records word for word, hence the values --; and the
    callbacks themselves sit on the analytic values.  (The shards' launches leave compact block rows, the 8e7-chain handle's
    lanes see 122 summands and leave wide ones: both row forms meet here.)"""
    from montecarlo_amd import sharding
    W, M_SHARD, sweeps, cb_every = 8, M_FULL, 30, 10
    M_GLOBAL = W * M_SHARD
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
    shards = []
    for r in range(W):
        start, stop = sharding.shard_range(M_GLOBAL, r, W)
        assert (start, stop) == (r * M_SHARD, (r + 1) * M_SHARD)
        e = gpu.HipEngine(n_chains=stop - start, chain_offset=start, n_chains_global=M_GLOBAL, **kw)
        e.init_uniform(-2.0, 2.0)
        shards.append(e)
    big = gpu.HipEngine(n_chains=M_GLOBAL, **kw)
    big.init_uniform(-2.0, 2.0)
    rows_sharded, rows_big = [], []
    for t in range(1, sweeps + 1):
        if t % cb_every == 0:
            for e in shards:
                e.sweep_reduce_begin(1)                   # sums formed inside the sweep launch, as in bench.py
            parts = [e.reduce_end_exact() for e in shards]
            merged = parts[0][0].copy()
            for rec, steps in parts[1:]:
                assert steps == parts[0][1]
                merged = gpu.xsum_merge(merged, rec)      # integers: any order gives the same words
            rows_sharded.append((merged, shards[0].reduce_records_value(merged, parts[0][1])))
            big.sweep_reduce_begin(1)
            rec, steps = big.reduce_end_exact()
            assert steps == parts[0][1] == t
            rows_big.append((rec, big.reduce_records_value(rec, steps)))
        else:
            for e in shards:
                e.sweep(1)
            big.sweep(1)
    for (got_rec, got), (want_rec, want) in zip(rows_sharded, rows_big):
        assert got[3] == want[3] == M_GLOBAL
        assert np.array_equal(got_rec, want_rec), [i for i in range(got_rec.shape[0]) if not np.array_equal(got_rec[i], want_rec[i])]
        assert np.array_equal(bits(got), bits(want))          # what DESIGN section 3.8 promises: the same bits for 8 shards and for one
    # the callbacks' values 30 sweeps after U(-2,2): <e> on its way down from 4/3 to 1/(2 beta), the acceptance of the
    # outlying chains (0.7 at |x| = 2) still pulls the cumulative ratio below its equilibrium 0.9365
    assert 0.85 < rows_sharded[-1][1][4] / M_GLOBAL < 0.94 and 0.25 < rows_sharded[-1][1][0] / M_GLOBAL < 4.0 / 3.0
    acc_total = sum(int(e.counter_totals()[0][0]) for e in shards)
    assert acc_total == int(big.counter_totals()[0][0])
    for r, e in enumerate(shards):
        for off in (0, 5_000_000 - 2048, M_SHARD - 4096):
            got = e.download_strided(off, 1, 4096)
            assert np.array_equal(bits(got), bits(big.download_strided(r * M_SHARD + off, 1, 4096)))
            o = oracle.OracleSim(4096, chain_offset=r * M_SHARD + off, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1)
            o.init_uniform(-2.0, 2.0)
            o.make_steps(sweeps, threads=4)
            assert np.array_equal(bits(got), bits(o.state()[0])), (r, off)
            o.close()
    for e in shards + [big]:
        e.close()


def test_config5_pgmc_at_1e7_chains_through_pgmc_steps(gpu, oracle):
    """BASELINE config 5 at its full size: PolicyGuided MC on particle_1d, M = 1e7 chains, pool sigma = (0.2, 0.1),
    w = (0.6, 0.4), optimisers (Static, VPG), q_batch_size = 1, sampler + estimator + update every time step
    (PGMC_harmonic_oscillator.jl:14-33; src/PolicyGuided/estimator.jl:111-134), issued through amc_pgmc_steps (one fused
    sweep + estimator launch per step, learning step in the launch's tail).
      1. The first 3 time steps against a FULL 1e7-chain oracle run on its own: its sweep, its GradientData fold (the same
         integer sum as the device's: reproducible sums), its learning_step! -- nothing fed back.  After
         every step the learned sigma is EQUAL, after the third every position and every per-chain counter is.
      2. 13 more steps one call at a time, followed by oracle runs of three 4096-chain slices (same global chain ids); a
         slice cannot learn by itself (the gradient is a sum over all 1e7 chains), so it takes the device's sigma: positions
         and per-chain counters bit for bit after every step.
      3. 400 more steps in one call: sigma_2 -> 1.2 +- 0.2, sigma_1 stays 0.2 exactly, <e> = 0.25 (pgmc_test.jl:45,50)."""
    eta, n_follow, n_full = 0.5, 16, 3
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=42)
    e = gpu.HipEngine(n_chains=M_FULL, **kw)
    e.init_uniform(-2.0, 2.0)
    full = oracle.OracleSim(M_FULL, **kw)
    full.init_uniform(-2.0, 2.0)
    sig_dev = []
    for t in range(n_full):
        e.pgmc_steps(1, [1], 1, [1], [eta], [0.0])          # optimiser id 1 = VPG
        full.make_steps(1, threads=16)
        gd = full.pg_estimate([1], 1)[0]
        full.set_sigma(1, oracle.learning_step("VPG", eta, 0.0, full.get_sigma(1), list(gd[:4] / gd[4])))
        sig_dev.append(float(e.get_parameters(1)[0]))
        assert sig_dev[-1] == full.get_sigma(1), (t, sig_dev[-1], full.get_sigma(1))
        assert e.get_parameters(0)[0] == 0.2
    x_full = full.state()[0]
    assert np.array_equal(bits(e.download_state(want_e=False)[0]), bits(x_full))
    acc, tot = e.download_counters()
    ao, to = full.counters()
    assert np.array_equal(acc, ao) and np.array_equal(tot, to)
    del acc, tot, ao, to
    offs = (0, 4_999_998, M_FULL - 4096)
    slices = []
    for off in offs:
        o = oracle.OracleSim(4096, chain_offset=off, **kw)
        o.set_x(x_full[off:off + 4096])
        o.step = full.step
        o.lib.amo_set_estimator_step(o.h, n_full)
        o.set_sigma(1, sig_dev[-1])
        a_s, t_s = full.counters()
        p = __import__("ctypes").POINTER(__import__("ctypes").c_int64)
        a_s = np.ascontiguousarray(a_s[:, off:off + 4096]); t_s = np.ascontiguousarray(t_s[:, off:off + 4096])
        o.lib.amo_set_counters(o.h, a_s.ctypes.data_as(p), t_s.ctypes.data_as(p))
        slices.append(o)
    full.close()
    del x_full
    for t in range(n_full, n_follow):
        e.pgmc_steps(1, [1], 1, [1], [eta], [0.0])
        s1 = float(e.get_parameters(1)[0])
        sig_dev.append(s1)
        for o, off in zip(slices, offs):
            o.make_steps(1)
            o.pg_estimate([1], 1)                           # moves x to (x + d) - d like the reference
            o.set_sigma(1, s1)
            assert np.array_equal(bits(e.download_strided(off, 1, 4096)), bits(o.state()[0])), (t, off)
    assert sig_dev[0] != 0.1 and all(b > a for a, b in zip(sig_dev, sig_dev[1:]))      # sigma grows towards 1.2
    acc, tot = e.download_counters()
    for o, off in zip(slices, offs):
        ao, to = o.counters()
        assert np.array_equal(acc[:, off:off + 4096], ao) and np.array_equal(tot[:, off:off + 4096], to)
        o.close()
    del acc, tot
    # the last ten of the 400 with the callback sums formed in the tenth launch (amc_pgmc_steps_reduce_begin) and ten more
    # queued behind them before the sums are read -- the pipelined form of config 5's callbacks every 10
    e.pgmc_steps(390, [1], 1, [1], [eta], [0.0])
    e.pgmc_steps(10, [1], 1, [1], [eta], [0.0], reduce_begin=True)
    x_then = e.download_strided(0, 997, 10_000)
    e.pgmc_steps(10, [1], 1, [1], [eta], [0.0])
    red = e.reduce_end()
    assert e.get_parameters(0)[0] == 0.2
    assert float(e.get_parameters(1)[0]) == pytest.approx(KATS["pgmc"]["sigma_star"], abs=KATS["pgmc"]["sigma_atol"])
    assert red[3] == M_FULL and red[0] / M_FULL == pytest.approx(0.25, abs=2e-3)
    assert np.mean(x_then ** 2) == pytest.approx(red[0] / M_FULL, abs=2e-2)            # the sums are those of the state at t = 416
    acc_tot = e.counter_totals()
    # callback_acceptance: mean of per-chain ratios (from the fold of the step log) vs the ratio of the pool totals
    assert red[4:] / M_FULL == pytest.approx(acc_tot[0] / acc_tot[1], abs=1e-2)
    red = e.reduce()
    assert np.all(e.pg_get_accumulated([1])[:, 4] == 0)     # every step ended in an update: accumulators are empty
    e.close()


def test_pgmc_example_learning_curve_on_device(gpu):
    """The reference's published PGMC learning curve (learning.png of PGMC_harmonic_oscillator.jl):
    VPG eta = 1e-3 takes sigma_2 from 0.1 to ~0.33 at t = 1e3.  Same configuration on the device with 2e5 chains (the
    gradient noise of the M = 10 example averaged out): the curve's value at t = 1e3 and its monotone rise."""
    from test_oracle_reference_tests import _pgmc_example_sigma_at
    s = _pgmc_example_sigma_at(None, 200_000, 42, [250, 500, 1000])
    assert s[0] < s[1] < s[2]
    assert s[2] == pytest.approx(0.33, abs=0.03)


def test_accept_filter_soak_at_full_size(gpu, monkeypatch):
    """1e7 chains x 2000 steps = 2e10 accept decisions, once through the filter and once through the reference-ordered
    arithmetic alone (AMC_EXACT_ACCEPT=1): every position and the accepted total must be identical.  About 2.4e6 of the
    filtered run's decisions fall into the filter's interval and are settled by the exact arithmetic there too."""
    kw = dict(n_chains=M_FULL, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=5, per_chain_counters=False)
    runs = []
    for exact in ("0", "1"):
        monkeypatch.setenv("AMC_EXACT_ACCEPT", exact)
        e = gpu.HipEngine(**kw)
        e.init_uniform(-2, 2)
        e.sweep(1)
        e.sweep(999)
        mid = int(e.counter_totals()[0][0])
        e.sweep(1000)
        runs.append((e.download_state(want_e=False)[0], int(e.counter_totals()[0][0]), mid))
        e.close()
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
    assert np.array_equal(bits(runs[0][0]), bits(runs[1][0]))
    rate = (runs[0][1] - runs[0][2]) / (1000 * M_FULL)              # the second thousand steps: equilibrated
    assert rate == pytest.approx(KATS["analytic"]["acceptance"]["beta=2.0,sigma=0.1"], abs=1e-4)


def test_pick_table_and_filter_soak_mixed_pool_at_full_size(gpu, monkeypatch):
    """The K = 2 counterpart: 1e7 chains x 1000 steps of the double well with weights (0.3, 0.7) -- the cumulative weight
    0.3 is no multiple of 2^-12, so one cell of the pick table is open and ~2.4e-4 of the picks (one wave-step in 32) take
    the accept draw's 24 low bits and the full categorical walk -- once on the common path (12-bit pick table, accept
    filter) and once with every wave sent through the walk and the reference-ordered accept (AMC_EXACT_ACCEPT=1): positions
    and every per-chain, per-move counter must be identical."""
    kw = dict(n_chains=M_FULL, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.3, 0.7], seed=11)
    runs = []
    for exact in ("0", "1"):
        monkeypatch.setenv("AMC_EXACT_ACCEPT", exact)
        e = gpu.HipEngine(**kw)
        e.init_uniform(-2, 2)
        for _ in range(100):
            e.sweep(1)                   # single-step launches (pre-formed draws) ...
        e.sweep(900)                     # ... and the multi-step form
        acc, tot = e.download_counters()
        runs.append((e.download_state(want_e=False)[0], acc, tot))
        e.close()
    assert np.array_equal(bits(runs[0][0]), bits(runs[1][0]))
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    tot = runs[0][2]
    assert np.all(tot.sum(axis=0) == 1000)
    assert tot[0].sum() / (1000 * M_FULL) == pytest.approx(0.3, abs=2e-5)       # 1e10 picks: sd 4.6e-6


def test_accept_filter_soak_script_proposal_at_full_size(gpu, monkeypatch):
    """The script-defined counterpart (round 5: accept_filter_arg): 1e7 chains x 1000 steps of a Langevin proposal written as
    expressions -- its arg = (dlogp + logq_b) - logq_f has nothing that cancels -- = 1e10 decisions, once through the filter and
    once through the reference-ordered decision alone (AMC_EXACT_ACCEPT=1): every position and the accepted total identical,
    and the acceptance where MALA's is for U = x^2, beta = 2, sigma = 0.4."""
    sample = "-2.0*sigma*sigma*x + sigma*z"
    logq = "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)"
    kw = dict(n_chains=M_FULL, potential="harmonic", beta=2.0, sigma=[0.4], weight=[1.0], seed=17, per_chain_counters=False,
              proposal=(sample, logq, None))
    runs = []
    for exact in ("0", "1"):
        monkeypatch.setenv("AMC_EXACT_ACCEPT", exact)
        e = gpu.HipEngine(**kw)
        e.init_uniform(-2, 2)
        for _ in range(10):
            e.sweep(1)
        e.sweep(990)
        runs.append((e.download_state(want_e=False)[0], int(e.counter_totals()[0][0])))
        e.close()
    assert runs[0][1] == runs[1][1]
    assert np.array_equal(bits(runs[0][0]), bits(runs[1][0]))
    assert 0.8 < runs[0][1] / (1000 * M_FULL) < 1.0          # a Langevin step of this size is accepted most of the time


@pytest.mark.parametrize("group", ["nccl", "store"])
def test_sharded_pgmc_device_resident_over_rccl(group):
    """PGMC with the shards connected: PolicyGradientEstimator.connect_shards() hands the engines a communicator of
    their own (the ncclUniqueId travels over a torch.distributed NCCL process group, or over the launcher's TCP store
    alone -- sharding.init_store_group, no process group), the fold is all-reduced in place on the device and the
    learning step stays there; the learned sigma equals the host path's (pg_estimate + host-side sum).  One rank here
    (the GPU box has one GPU); the code path is the N-rank one."""
    import json, subprocess, sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541" if group == "nccl" else "29543", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", AMC_TEST_GROUP=group)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux", "pgmc_comm_worker.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["comm"]["connected"] and out["comm"]["device_resident"] and not out["host"]["device_resident"]
    assert out["comm"]["sigma"][0] == out["host"]["sigma"][0] == 0.2
    # the fold is a reproducible sum: device-resident + in-place all-reduce and host path agree to the bit
    assert out["comm"]["sigma"][1] == out["host"]["sigma"][1] and out["comm"]["sigma"][1] > 0.5
    assert out["comm"]["x0"] == out["host"]["x0"]
    _check_comm_rows(out, 1)


def _check_comm_rows(out, world):
    """Callbacks interleaved with the grouped PGMC time steps: their all-reduce runs on the engine's communication stream,
    the estimator's on its main stream, both on one communicator -- same rows as the host path, and the communicator
    itself reports the ranks it spans."""
    assert out["comm"]["comm"]["n_ranks"] == world and "librccl" in out["comm"]["comm"]["librccl"]
    assert [t for t, _ in out["comm"]["energy"]] == [t for t, _ in out["host"]["energy"]] == [0] + list(range(10, 121, 10))
    assert [v for _, v in out["comm"]["energy"]] == [v for _, v in out["host"]["energy"]]
    assert np.array_equal(np.array([v for _, v in out["comm"]["acceptance"]]), np.array([v for _, v in out["host"]["acceptance"]]),
                          equal_nan=True)


def test_two_gpus_callbacks_interleaved_with_pgmc_steps(gpu):
    """The same worker on TWO ranks, one GPU each, under the driver's launcher: RCCL over xGMI carries the estimator's
    in-place all-reduce (engine stream) and the callbacks' sums (communication stream) of one communicator.  Needs two
    devices: skipped on the one-GPU boxes this repository's own runs get (the N > 1 path has never run on hardware
    here); on a multi-GPU node it is the hardware check of that path."""
    import json, socket, subprocess, sys
    if gpu.device_count() < 2:
        pytest.skip("needs two GPUs")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AMC_TEST_GROUP="store")
    cmd = ["timeout", "-k", "10", "600", sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux", "pgmc_comm_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["comm"]["world"] == 2 and out["comm"]["connected"] and out["comm"]["device_resident"]
    assert out["comm"]["sigma"][0] == out["host"]["sigma"][0] == 0.2
    assert out["comm"]["sigma"][1] == out["host"]["sigma"][1] and out["comm"]["sigma"][1] > 0.5      # two GPUs, one GPU's bits
    _check_comm_rows(out, 2)


@pytest.mark.parametrize("case", ["harmonic_K1", "double_well_K2", "harmonic_f32"])
def test_stationary_density_goodness_of_fit_at_full_size(gpu, case):
    """The statistic behind test/distribution_test.jl:33-37 (pooled positions against the target), as sharp as 1e7
    INDEPENDENT chains allow: after burn-in the positions at one time are 1e7 independent draws from the target, so a
    200-bin histogram (amc_histogram, device side) must fit exp(-beta U) with chi^2 / dof = 1 +- a few sqrt(2 / dof).
    The reference's own tolerance (1e-3 on mean and std of ~1e7 correlated samples) sees a bias of 1e-3; this sees a relative
    density error of ~3e-3 in ANY bin, i.e. anything the 12-bit accept brackets, the 28-bit Box-Muller angle, the float
    accept filter or the table-driven exp / log could have bent.  Harmonic K = 1 (configs 2 / 4), double well K = 2
    (config 3), and the Float32 state type."""
    from scipy import integrate, stats
    beta, n_bins = 2.0, 200
    if case == "double_well_K2":
        kw = dict(potential="double_well", sigma=[0.1, 1.0], weight=[0.5, 0.5])
        lo, hi, burn = -2.0, 2.0, 3000
        logp = lambda x: -beta * (x * x - 1.0) ** 2
    else:
        kw = dict(potential="harmonic", sigma=[0.1], weight=[1.0], per_chain_counters=False)
        lo, hi, burn = -2.0, 2.0, 6000                      # integrated autocorrelation ~80 sweeps at sigma = 0.1: 75 of them
        logp = lambda x: -beta * x * x
        if case == "harmonic_f32":
            kw.update(dtype="f32")
    e = gpu.HipEngine(n_chains=M_FULL, beta=beta, seed=20260304, **kw)
    e.init_uniform(-2, 2)
    e.sweep(burn)
    counts = e.histogram(lo, hi, n_bins).astype(np.float64)
    assert counts[n_bins + 2] == 0 and counts.sum() == M_FULL          # no NaN; every chain counted once
    edges = np.linspace(lo, hi, n_bins + 1)
    z = integrate.quad(lambda x: np.exp(logp(x)), -8, 8, epsabs=1e-14, epsrel=1e-13)[0]
    if case == "double_well_K2":
        p_in = np.array([integrate.quad(lambda x: np.exp(logp(x)), a, b, epsabs=1e-15, epsrel=1e-12)[0] for a, b in zip(edges[:-1], edges[1:])]) / z
        p_lo = integrate.quad(lambda x: np.exp(logp(x)), -8, lo, epsabs=1e-16)[0] / z
    else:
        cdf = stats.norm(0.0, 1.0 / np.sqrt(2 * beta)).cdf
        p_in = np.diff(cdf(edges))
        p_lo = cdf(lo)
    p = np.concatenate([p_in, [p_lo, p_lo]])                           # symmetric targets: the two tails are equal
    expect = p * M_FULL
    obs = counts[:n_bins + 2]
    # Float32 positions sit on a grid; bin edges at multiples of 0.02 are not grid points in general, fine at this bin width
    keep = expect >= 20
    chi2 = float(np.sum((obs[keep] - expect[keep]) ** 2 / expect[keep]))
    dof = int(keep.sum()) - 1
    assert dof >= 150
    assert abs(chi2 - dof) < 5.0 * np.sqrt(2.0 * dof), (case, chi2, dof)    # 5 sigma of the chi^2 distribution
    # the same numbers as a mean / variance statement, tighter than the reference's 1e-3
    r = e.reduce()
    assert r[1] / M_FULL == pytest.approx(0.0, abs=6e-4)
    e.close()


@pytest.mark.parametrize("sigma", [0.6, 1.2, 2.0])
def test_policy_gradient_sums_against_quadrature_at_full_size(gpu, sigma):
    """What test/pgmc_test.jl pins loosely (sigma -> 1.2 +- 0.2), taken apart and pinned to four digits: with the chains at
    stationarity (x ~ N(0, 1/(2 beta))) the estimator's sums over 1e7 chains are Monte-Carlo estimates of closed-form
    integrals (gradients.jl:93-109 with the model's reward delta^2, particle_1d.jl:42-44):
        j / n            -> J(sigma) = E[delta^2 min(1, exp(dlogp))]           (the objective the optimisers climb)
        grad_j / n       -> dJ/dsigma                                          (score-function identity)
        grad_logq / n    -> 0,    g / n -> 2 / sigma^2                         (Fisher information of the Gaussian scale)
    J and dJ/dsigma by a 2-D quadrature on the host; tolerances are 6 standard errors of 1e7 samples."""
    beta = 2.0
    e = gpu.HipEngine(n_chains=M_FULL, potential="harmonic", beta=beta, sigma=[sigma], weight=[1.0], seed=77, per_chain_counters=False)
    e.init_uniform(-2, 2)
    e.sweep(400)                                              # steps of this size decorrelate in a few sweeps
    g = e.pg_estimate([0], 1)[0]
    n = g[4]
    assert n == M_FULL

    def J(s):
        x = np.linspace(-3.5, 3.5, 2801)[:, None]             # 7 sigma_x; exact to ~1e-9 on this grid but for the kink of min()
        d = np.linspace(-8.0 * s, 8.0 * s, 6401)[None, :]
        px = np.exp(-beta * x * x)
        px = px / np.trapezoid(px[:, 0], x[:, 0])
        qd = np.exp(-d * d / (2 * s * s)) / np.sqrt(2 * np.pi * s * s)
        alpha = np.minimum(1.0, np.exp(-beta * ((x + d) ** 2 - x * x)))
        inner = np.trapezoid(qd * d * d * alpha, d[0], axis=1)
        return float(np.trapezoid(px[:, 0] * inner, x[:, 0]))

    j_true = J(sigma)
    dj_true = (J(sigma + 1e-3) - J(sigma - 1e-3)) / 2e-3
    se_j = 1.5 * sigma ** 2 / np.sqrt(n)                       # sd(delta^2 alpha) < sd(delta^2) = sqrt(2) sigma^2
    se_dj = 4.0 * sigma / np.sqrt(n)                           # sd(j grad_logq) ~ a few sigma
    assert g[0] / n == pytest.approx(j_true, abs=6 * se_j + 2e-5)
    assert g[1] / n == pytest.approx(dj_true, abs=6 * se_dj + 1e-4)
    assert g[2] / n == pytest.approx(0.0, abs=6 * np.sqrt(2.0) / sigma / np.sqrt(n))
    assert g[3] / n == pytest.approx(2.0 / sigma ** 2, rel=6 * np.sqrt(28.0) / 2.0 / np.sqrt(n))    # var of (z^2 - 1)^2 = 60 - 4 -> sd/mean = sqrt(56)/2
    if sigma == 1.2:
        assert abs(dj_true) < 5e-3 and j_true == pytest.approx(0.18597, abs=2e-4)      # the optimum the reference's test looks for
    # the SWEEP's own dynamics against the same integral: one mc_step! moves a chain by delta with probability alpha, so the
    # mean squared displacement of one sweep is E[delta^2 alpha] = J(sigma) too (and ties the estimator's objective to what the
    # sampler really does); 1e6 chains sampled before and after one make_step!
    m = 1_000_000
    before = e.download_strided(0, 10, m)
    acc0, tot0 = e.counter_totals()
    e.sweep(1)
    after = e.download_strided(0, 10, m)
    acc1, tot1 = e.counter_totals()
    msd = float(np.mean((after - before) ** 2))
    assert msd == pytest.approx(j_true, abs=6 * 1.5 * sigma ** 2 / np.sqrt(m) + 2e-5)
    # accepted <=> moved by more than rounding: a rejected chain is put back by re-adding the negated action, x = (x + d) + (-d),
    # which differs from x in the last bits for a good part of the rejects (metropolis.jl:187)
    moved = float(np.mean(np.abs(after - before) > 1e-9))
    assert moved == pytest.approx((acc1[0] - acc0[0]) / (tot1[0] - tot0[0]), abs=6 * 0.5 / np.sqrt(m))
    assert 0.05 < float(np.mean((after != before) & (np.abs(after - before) <= 1e-9))) < 0.5      # the quirk is there
    e.close()


@pytest.mark.parametrize("derivative", ["given", "differentiated"])
def test_two_parameter_gradient_sums_against_quadrature_at_full_size(gpu, derivative):
    """(`differentiated`, round 6: the handle is created WITHOUT the partial derivatives -- the engine differentiates logq itself,
    dual numbers in the kernel, DESIGN.md section 3.11 -- and must meet the same integrals.)
    The same for a policy with TWO parameters, delta = mu + sigma z (amc_create_vector_policy_model): the proposal is not
    symmetric, log q_b - log q_f = -2 delta mu / sigma^2 enters the acceptance, and with the chains at stationarity
        j / n           -> J(mu, sigma) = E[delta^2 min(1, exp(dlogp + logq_b - logq_f))]
        grad_j / n      -> (dJ/dmu, dJ/dsigma)                 (score-function identity, gradients.jl:106)
        grad_logq / n   -> (0, 0),   g / n -> diag(1, 2) / sigma^2   (Fisher information of a Gaussian's mean and scale)
    J by a 2-D quadrature on the host, its gradient by central differences; tolerances are 6 standard errors of 1e7 samples."""
    beta, mu, sigma = 2.0, 0.15, 0.9
    sample = "theta0 + theta1*z"
    logq = "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)"
    dlogq = ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"]
    e = gpu.HipEngine(n_chains=M_FULL, potential="harmonic", beta=beta, sigma=[[mu, sigma]], weight=[1.0], seed=78,
                      per_chain_counters=False, proposal=(sample, logq, dlogq if derivative == "given" else None), n_params=2)
    e.init_uniform(-2, 2)
    e.sweep(300)
    x = e.download_strided(0, 10, 1_000_000)
    assert np.mean(x) == pytest.approx(0.0, abs=6 * 0.5 / 1e3) and np.mean(x * x) == pytest.approx(1 / (2 * beta), abs=3e-3)   # still the target
    g = e.pg_estimate([0], 1)[0]
    n = g[-1]
    assert g.shape == (10,) and n == M_FULL

    def J(m, s):
        xx = np.linspace(-3.5, 3.5, 2801)[:, None]
        d = np.linspace(m - 8.0 * s, m + 8.0 * s, 6401)[None, :]
        px = np.exp(-beta * xx * xx)
        px = px / np.trapezoid(px[:, 0], xx[:, 0])
        qd = np.exp(-(d - m) ** 2 / (2 * s * s)) / np.sqrt(2 * np.pi * s * s)
        alpha = np.minimum(1.0, np.exp(-beta * ((xx + d) ** 2 - xx * xx) - 2.0 * d * m / (s * s)))
        inner = np.trapezoid(qd * d * d * alpha, d[0], axis=1)
        return float(np.trapezoid(px[:, 0] * inner, xx[:, 0]))

    j_true = J(mu, sigma)
    dj_mu = (J(mu + 1e-3, sigma) - J(mu - 1e-3, sigma)) / 2e-3
    dj_sg = (J(mu, sigma + 1e-3) - J(mu, sigma - 1e-3)) / 2e-3
    se = 1.0 / np.sqrt(n)
    assert g[0] / n == pytest.approx(j_true, abs=6 * 1.5 * (sigma ** 2 + mu ** 2) * se + 2e-5)
    assert g[1] / n == pytest.approx(dj_mu, abs=6 * 3.0 * se + 1e-4)         # sd(j dlogq/dmu) ~ sigma |z|^3 / sigma: a few units
    assert g[2] / n == pytest.approx(dj_sg, abs=6 * 4.0 * sigma * se + 1e-4)
    assert abs(dj_mu) > 5 * 6 * 3.0 * se                                      # the gradient along the drift is many tolerances from zero: a real check
    assert g[3] / n == pytest.approx(0.0, abs=6 / sigma * se) and g[4] / n == pytest.approx(0.0, abs=6 * np.sqrt(2.0) / sigma * se)
    G = g[5:9].reshape(2, 2) / n
    assert G[0, 0] == pytest.approx(1 / sigma ** 2, rel=6 * np.sqrt(2.0) * se)            # var(z^2) = 2
    assert G[1, 1] == pytest.approx(2 / sigma ** 2, rel=6 * np.sqrt(28.0) / 2.0 * se)
    assert G[0, 1] == G[1, 0] and G[0, 1] == pytest.approx(0.0, abs=6 * np.sqrt(10.0) / sigma ** 2 * se)   # E[z (z^2-1)] = 0, var = 10
    e.close()


def test_class_pool_one_launch_per_time_step_at_full_size(gpu):
    """Round 6: a pool that mixes policy types (Gaussian + Langevin, the Langevin class WITHOUT its derivative: the engine
    differentiates logq) takes its whole PGMC time step in ONE launch (every learnable move, the sweep in front: the route of every
    other pool; reference: one estimator pass over all learnable moves, estimator.jl:111-134).  At 1e7 chains -- the full grid,
    the tail's two ticket levels -- three time steps equal the one-launch-per-move route (AMC_CLASS_PER_MOVE=1) bit for bit:
    parameters, every chain, every counter; and with AMC_NO_COLUMN_SKIP=1 (every GradientData column summed) likewise."""
    gauss = ("sigma*z", "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0",
             "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma")
    mala = ("-2.0*sigma*sigma*x + sigma*z", "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)", None)
    out = []
    for env in ({}, {"AMC_CLASS_PER_MOVE": "1"}, {"AMC_NO_COLUMN_SKIP": "1"}):
        os.environ.update(env)
        try:
            e = gpu.HipEngine(n_chains=M_FULL, potential="harmonic", beta=2.0, sigma=[0.3, 0.4], weight=[0.5, 0.5], seed=42,
                              classes=[gauss, mala], class_of_move=[0, 1])
        finally:
            for k in env:
                del os.environ[k]
        e.init_uniform(-2, 2)
        assert e.pg_route_code(2, 1, fused=True)[0] == (0 if "AMC_CLASS_PER_MOVE" in env else 2)
        e.pgmc_steps(3, [0, 1], 1, [1, 2], [1e-3, 1e-3], [0.0, 0.0])              # VPG and BLPG
        x = e.download_state()[0]
        acc, tot = e.counter_totals()
        out.append(([e.get_parameters(k)[0] for k in range(2)], int(np.bitwise_xor.reduce(x.view(np.uint64))), float(x[12345]),
                    acc.tolist(), tot.tolist()))
        e.close()
    assert out[0] == out[1] == out[2]
    assert out[0][0][0] != 0.3 and out[0][0][1] != 0.4 and sum(out[0][4]) == 3 * M_FULL


def test_launch_plans_of_the_round_at_full_size(gpu):
    """Round 6, at 1e7 chains (the full grid, 13 trips per lane, the tail's two ticket levels): (a) the pool of the reference's
    test/pgmc_test.jl:16-27 -- seven moves, six optimisers, q_batch_size = 10 -- whose estimator call goes in launches of four moves;
    (b) two learnable moves of a two-parameter policy as a chain of launches with their own tails.  Each against the route it
    replaced (AMC_NP_SMALL_LAUNCHES=1): parameters, every chain, every counter the same bits; and the learning goes the reference's
    way (test/pgmc_test.jl:50: sigma grows from 0.1 towards 1.2 under every optimiser)."""
    drift = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)", None)
    cases = [
        (dict(potential="harmonic", beta=2.0, sigma=[0.1] * 7, weight=[0.4] + [0.1] * 6, seed=42),
         ([1, 2, 3, 4, 5, 6], 10, [1, 2, 3, 4, 5, 6], [0.001, 0.001, 1e-6, 1e-2, 1e-6, 1e-6], [0.0, 0.0, 1e-6, 1e-6, 1e-6, 1e-6]), 2),
        (dict(potential="harmonic", beta=2.0, sigma=[[0.0, 0.5], [0.1, 0.9]], weight=[0.5, 0.5], seed=42, proposal=drift, n_params=2),
         ([0, 1], 1, [1, 4], [1e-3, 5e-3], [0.0, 1e-6]), 3),
    ]
    for kw, (learn, q, kinds, h0, h1), steps in cases:
        out = []
        for env in ({}, {"AMC_NP_SMALL_LAUNCHES": "1"}):
            os.environ.update(env)
            try:
                e = gpu.HipEngine(n_chains=M_FULL, **kw)
            finally:
                for k in env:
                    del os.environ[k]
            e.init_uniform(-2, 2)
            e.pgmc_steps(steps, learn, q, kinds, h0, h1)
            x = e.download_state()[0]
            acc, tot = e.counter_totals()
            out.append(([list(e.get_parameters(k)) for k in range(e.n_moves)], int(np.bitwise_xor.reduce(x.view(np.uint64))), float(x[7654321]),
                        acc.tolist(), tot.tolist()))
            e.close()
        assert out[0] == out[1], kw["sigma"]
        assert sum(out[0][4]) == steps * M_FULL
        if len(learn) == 6:
            sig = [p[0] for p in out[0][0]]
            assert sig[0] == 0.1 and all(s > 0.1 for s in sig[1:]), sig
