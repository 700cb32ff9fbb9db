"""The reference's own tests for this path, run on the CPU oracle with the reference's tolerances.

These are what "pin" the oracle: the reference holds no bit-level vectors, only
  test/distribution_test.jl:9-39   mean / std of sampled positions, atol 1e-3
  test/pgmc_test.jl:10-52          <e> = 0.25 +- 0.05, learned sigma = 1.2 +- 0.2, Static untouched
CPU only; the pgmc test also exercises the package's host logic through the engine_factory seam.
"""
import json
import math
import os
import tempfile

import numpy as np
import pytest

import montecarlo_amd as ma

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
    KATS = json.load(f)


@pytest.mark.slow
@pytest.mark.parametrize("beta", KATS["distribution"]["betas"])
def test_harmonic_oscillator_distribution(oracle, beta):
    """test/distribution_test.jl: sigma = 0.1, seed 42, steps 1e6, burn 1000, samples every 10 sweeps,
    positions of all chains pooled; mean ~ 0 and std ~ 1/sqrt(2 beta), both atol 1e-3.

    The reference uses M = 100, for which 1e-3 is only ~1.5 standard errors of the pooled mean
    (integrated autocorrelation ~80 sweeps); M = 500 makes the SAME tolerance a >3-sigma test, so
    a failure means a wrong sampler, not an unlucky seed."""
    k = KATS["distribution"]
    o = oracle.OracleSim(500, potential="harmonic", beta=beta, sigma=[k["sigma"]], weight=[1.0], seed=k["seed"])
    o.init_uniform(-2.0, 2.0)                      # chains = [System(4rand(rng) - 2, beta) ...]
    n, sx, sxx, mean_e = o.run_pooled_moments(k["steps"], k["burn"], k["block"][-1], threads=oracle.load().amo_max_threads())
    mean = sx / n
    std = math.sqrt(sxx / n - mean * mean)
    assert mean == pytest.approx(0.0, abs=k["atol"])
    assert std == pytest.approx(1 / math.sqrt(2 * beta), abs=k["atol"])
    # analytic companions: <e> = 1/(2 beta), acceptance = (2/pi) atan(2 s / sigma)
    assert mean_e == pytest.approx(KATS["analytic"]["mean_energy"][str(beta)], abs=1e-3)
    assert o.acceptance()[0] == pytest.approx(KATS["analytic"]["acceptance"][f"beta={beta},sigma=0.1"], abs=1e-3)


@pytest.mark.slow
def test_displacement_optimisation_all_optimisers(oracle):
    """test/pgmc_test.jl: 7 identical moves sigma0 = 0.2, weights (0.4, 0.1 x 6), one optimiser each,
    q_batch_size 10, estimator every step, update every 2 steps after burn 1000, steps 1e5."""
    k = KATS["pgmc"]
    seed, beta, M, sigma0 = 42, k["beta"], k["M"], k["sigma0"]
    chains = ma.ParticleChains.uniform(M, beta, -2.0, 2.0)
    pool = tuple(ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": sigma0}, w) for w in k["weights"])
    optimisers = (ma.Static(), ma.VPG(0.001), ma.BLPG(0.001), ma.BLAPG(1e-6, 1e-6), ma.NPG(1e-2, 1e-6),
                  ma.ANPG(1e-6, 1e-6), ma.BLANPG(1e-6, 1e-6))
    steps, burn = k["steps"], k["burn"]
    sampletimes = ma.build_schedule(steps, burn, [0, 10])
    with tempfile.TemporaryDirectory() as path:
        algorithm_list = (
            dict(algorithm=ma.Metropolis, pool=pool, seed=seed, parallel=False, engine_factory=oracle.OracleEngine),
            dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=optimisers,
                 q_batch_size=k["q_batch_size"], parallel=True),
            dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,),
                 scheduler=ma.build_schedule(steps, burn, k["update_every"])),
            dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
                 scheduler=sampletimes),
            dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=sampletimes),
        )
        simulation = ma.Simulation(chains, algorithm_list, steps, path=path)
        ma.run(simulation)
        energies = np.loadtxt(os.path.join(path, "energy.dat"))[:, 1]
        assert energies.mean() == pytest.approx(k["mean_energy"], abs=k["energy_atol"])
        for i, opt in enumerate(optimisers):
            lines = open(os.path.join(path, "parameters", str(i + 1), "parameters.dat")).read().split("\n")
            last = float(lines[-2].split(" ", 1)[1].strip("[]"))
            if isinstance(opt, ma.Static):
                assert last == sigma0
            else:
                assert last == pytest.approx(k["sigma_star"], abs=k["sigma_atol"]), type(opt).__name__


def test_pgmc_objective_peaks_near_sigma_star(oracle):
    """J(sigma) = E[delta^2 alpha] at beta = 2 is maximal near sigma = 1.2
    (1.1 -> 0.18503, 1.2 -> 0.18597, 1.3 -> 0.18545); checks the estimator's j and the sign of grad j."""
    vals = {}
    for sigma in (0.6, 1.2, 2.0):
        o = oracle.OracleSim(20000, potential="harmonic", beta=2.0, sigma=[sigma], weight=[1.0], seed=3)
        x = np.random.default_rng(0).normal(0, 0.5, 20000)     # stationary ensemble N(0, 1/(2 beta))
        o.set_x(x)
        g = o.pg_estimate([0], 10)[0]
        vals[sigma] = (g[0] / g[4], g[1] / g[4])
    assert vals[1.2][0] == pytest.approx(0.18597, abs=3e-3)
    assert vals[0.6][0] < vals[1.2][0] > vals[2.0][0]
    assert vals[0.6][1] > 0 > vals[2.0][1]                      # gradient points towards sigma*


def _pgmc_example_sigma_at(engine_factory, M, seed, times):
    """example/particle_1d/harmonic_oscillator/PGMC_harmonic_oscillator.jl:9-35: pool sigma = (0.2, 0.1), weights
    (0.6, 0.4), optimisers (Static, VPG(0.001)), estimator and update every step; returns sigma_2 at `times`."""
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, 0.6),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.4))
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=seed, engine_factory=engine_factory),
          dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.001))),
          dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
          dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=list(times)))
    with tempfile.TemporaryDirectory() as path:
        ma.run(ma.Simulation(chains, al, times[-1], path=path))
        rows = open(os.path.join(path, "parameters", "2", "parameters.dat")).read().splitlines()
    assert pool[0].sigma == 0.2
    return [float(r.split("[")[1].strip("]")) for r in rows][1:]


def test_pgmc_example_learning_curve(oracle):
    """The reference publishes the learning curve of its PGMC example as a figure
    (example/particle_1d/harmonic_oscillator/learning.png, script PGMC_harmonic_oscillator.jl:45-51): with VPG
    eta = 1e-3, M = 10, sigma_2 has grown from 0.1 to ~0.33 at t = 1e3.  Eight seeds of the same
    configuration on the oracle."""
    s1000 = [_pgmc_example_sigma_at(oracle.OracleEngine, 10, seed, [1000])[0] for seed in range(40, 48)]
    assert np.mean(s1000) == pytest.approx(0.33, abs=0.03)
    assert all(abs(s - 0.33) < 0.08 for s in s1000)
