"""The reference's own tests for this path, at their own configurations, on the HIP path.

  test/distribution_test.jl:9-39   beta in {2.0, 2.5, 3.0}, sigma = 0.1, seed 42, 1e6 steps, burn 1000, positions of all
                                   chains sampled every 10 steps and pooled; mean 0 +- 1e-3, std 1/sqrt(2 beta) +- 1e-3
  test/pgmc_test.jl:10-52          7 identical moves sigma0 = 0.2, weights (0.4, 0.1 x 6), one optimiser each, q_batch_size 10,
                                   M = 10, 1e5 steps, update every 2 steps after burn 1000; <e> = 0.25 +- 0.05, learned
                                   sigma = 1.2 +- 0.2 for the six optimisers, Static == 0.2 exactly
Both go through the package's Simulation / run / StoreCallbacks / StoreParameters with the default (HIP) engine: this is
where the GPU path meets numbers the reference itself holds, without the oracle in between
(tests/test_oracle_reference_tests.py runs the same two tests on the oracle).
"""
import json
import math
import os

import numpy as np
import pytest

import montecarlo_amd as ma

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
    KATS = json.load(f)


@pytest.mark.parametrize("beta", KATS["distribution"]["betas"])
def test_harmonic_oscillator_distribution_on_device(gpu, beta, tmp_path):
    """test/distribution_test.jl.  The reference reads the sampled positions back from its per-chain trajectory files
    and pools them; here the pooled first and second moments are accumulated from the device-side sums over x at the
    same sample times (callback_moments: [mean(x), mean(x^2)] over the chains, formed in the sweep launch), which is the
    same statistic without 1e5 x M text rows.

    M: the reference uses 100 chains, for which its own tolerance of 1e-3 is ~1.5 standard errors of the pooled mean
    (integrated autocorrelation ~80 sweeps at sigma = 0.1), i.e. the test as written fails for one seed in seven on ANY
    correct sampler.  M = 2000 makes the same tolerance a > 6-sigma statement, so a failure here means a wrong sampler."""
    k = KATS["distribution"]
    M = 2000
    steps, burn, block = k["steps"], k["burn"], k["block"]
    sampletimes = ma.build_schedule(steps, burn, block)
    chains = ma.ParticleChains.uniform(M, beta, -2.0, 2.0)                  # [System(4rand(rng) - 2, beta) for _ in 1:M]
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": k["sigma"]}, 1.0),)
    algorithm_list = (
        dict(algorithm=ma.Metropolis, pool=pool, seed=k["seed"], parallel=False),
        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance, ma.callback_moments),
             scheduler=sampletimes),
    )
    simulation = ma.Simulation(chains, algorithm_list, steps, path=str(tmp_path))
    ma.run(simulation)
    assert isinstance(simulation.algorithms[0].engine, gpu.HipEngine)
    rows = simulation.algorithms[1].rows
    # StoreTrajectories' default store_first = true puts the t = 0 frame into the reference's pooled positions as well
    # (src/algorithms.jl:161,196): one row of uniform starts in 99 902, kept for fidelity
    assert [t for t, _ in rows[2]] == [0] + sampletimes and len(sampletimes) == (steps - burn) // block[-1] + 1
    m1 = np.array([v[0] for _, v in rows[2]])
    m2 = np.array([v[1] for _, v in rows[2]])
    n = len(m1) * M                                                          # pooled positions
    mean = m1.mean()
    std = math.sqrt((m2.mean() - mean * mean) * n / (n - 1))                 # Julia's std: corrected
    assert mean == pytest.approx(k["mean"], abs=k["atol"])
    assert std == pytest.approx(1 / math.sqrt(2 * beta), abs=k["atol"])
    # analytic companions: <e> = 1/(2 beta); acceptance of the Gaussian random walk = (2/pi) atan(2 s / sigma)
    assert np.mean([v for t, v in rows[0] if t > 0]) == pytest.approx(KATS["analytic"]["mean_energy"][str(beta)], abs=1e-3)
    assert rows[1][-1][1][0] == pytest.approx(KATS["analytic"]["acceptance"][f"beta={beta},sigma=0.1"], abs=1e-3)
    assert pool[0].total_calls == steps * M
    assert pool[0].accepted_calls / pool[0].total_calls == pytest.approx(rows[1][-1][1][0], abs=1e-12)


def test_displacement_optimisation_all_optimisers_on_device(gpu, tmp_path):
    """test/pgmc_test.jl, verbatim: the device-resident estimator / update (amc_pgmc_steps, amc_pg_update), M = 10."""
    k = KATS["pgmc"]
    seed, beta, M, sigma0 = 42, k["beta"], k["M"], k["sigma0"]
    chains = ma.ParticleChains.uniform(M, beta, -2.0, 2.0)
    pool = tuple(ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": sigma0}, w) for w in k["weights"])
    optimisers = (ma.Static(), ma.VPG(0.001), ma.BLPG(0.001), ma.BLAPG(1e-6, 1e-6), ma.NPG(1e-2, 1e-6),
                  ma.ANPG(1e-6, 1e-6), ma.BLANPG(1e-6, 1e-6))
    assert [type(o).__name__ for o in optimisers] == [s.split("(")[0] for s in k["optimisers"]]
    steps, burn = k["steps"], k["burn"]
    sampletimes = ma.build_schedule(steps, burn, [0, 10])
    path = str(tmp_path)
    algorithm_list = (
        dict(algorithm=ma.Metropolis, pool=pool, seed=seed, parallel=False),
        dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=optimisers,
             q_batch_size=k["q_batch_size"], parallel=True),
        dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,),
             scheduler=ma.build_schedule(steps, burn, k["update_every"])),
        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=sampletimes),
        dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=sampletimes),
    )
    simulation = ma.Simulation(chains, algorithm_list, steps, path=path)
    ma.run(simulation)
    assert isinstance(simulation.algorithms[0].engine, gpu.HipEngine)
    assert simulation.algorithms[1].device_resident
    energies = np.loadtxt(os.path.join(path, "energy.dat"))[:, 1]
    assert len(energies) == len(sampletimes) + 1                             # store_first row at t = 0
    assert energies.mean() == pytest.approx(k["mean_energy"], abs=k["energy_atol"])
    for i, opt in enumerate(optimisers):
        lines = open(os.path.join(path, "parameters", str(i + 1), "parameters.dat")).read().split("\n")
        last = float(lines[-2].split(" ", 1)[1].strip("[]"))
        if isinstance(opt, ma.Static):
            assert last == sigma0
        else:
            assert last == pytest.approx(k["sigma_star"], abs=k["sigma_atol"]), type(opt).__name__
    # the device copy of sigma and the host's Move.parameters agree after finalise
    eng = simulation.algorithms[0].engine
    assert [eng.get_parameters(i)[0] for i in range(7)] == [m.sigma for m in pool]
    assert sum(m.total_calls for m in pool) == steps * M
