#!/usr/bin/env python3
"""Policy-guided Monte Carlo on the harmonic oscillator: the width of one of two Gaussian displacement moves is
learned by policy gradient while the chains sample.

The driver the reference ships as example/particle_1d/harmonic_oscillator/PGMC_harmonic_oscillator.jl, on this
engine: pool sigma = (0.2, 0.1) with weights (0.6, 0.4), optimisers (Static, VPG(eta)), estimator and update at every
step, parameters stored with the callbacks.  With M chains the gradient estimate is ~M/10 times less noisy than in
the reference's M = 10 example, so eta may be raised accordingly (default: 0.001 * M / 10, capped at 0.5).

    python examples/pgmc_harmonic_oscillator.py [--chains 10] [--steps 100000] [--eta ETA]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlo_amd as ma   # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10 ** 5)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--beta", type=float, default=2.0)
    ap.add_argument("--eta", type=float, default=None)
    ap.add_argument("--path", default=None)
    ap.add_argument("--dtype", default="f64", choices=("f64", "f32"), help="Particle{T}: Float64 (reference scripts) or Float32")
    args = ap.parse_args(argv)

    seed, beta, M, steps = args.seed, args.beta, args.chains, args.steps
    eta = args.eta if args.eta is not None else min(0.5, 0.001 * M / 10)
    burn = min(1000, steps // 10)
    chains = ma.ParticleChains.uniform(M, beta, -2.0, 2.0, dtype=args.dtype)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, 0.6),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.4))
    optimisers = (ma.Static(), ma.VPG(eta))
    sampletimes = ma.build_schedule(steps, burn, [0, 10])
    path = args.path or f"data/PGMC/particle_1d/Harmonic/beta{beta}/M{M}/seed{seed}"

    algorithm_list = (
        dict(algorithm=ma.Metropolis, pool=pool, seed=seed),
        dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=optimisers),
        dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=sampletimes),
        dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=sampletimes),
        dict(algorithm=ma.PrintTimeSteps, scheduler=ma.build_schedule(steps, burn, max(1, steps // 10))),
    )
    simulation = ma.Simulation(chains, algorithm_list, steps, path=path, verbose=True)
    ma.run(simulation)

    rows = np.loadtxt(os.path.join(path, "energy.dat"), usecols=(0, 1))
    energies = rows[rows[:, 0] >= burn, 1]
    print(f"mean(energies), std(energies) = {energies.mean():.6f}, {energies.std():.6f}   (target <e> = {1 / (2 * beta):.6f})")
    prms = [ln.split(" ", 1) for ln in open(os.path.join(path, "parameters", "2", "parameters.dat")).read().splitlines()]
    t = np.array([int(a) for a, _ in prms])
    sigma = np.array([float(b.strip("[]")) for _, b in prms])
    for frac in (0.0, 0.1, 0.5, 1.0):
        i = min(len(t) - 1, int(frac * (len(t) - 1)))
        print(f"  sigma_2(t = {t[i]:>7d}) = {sigma[i]:.4f}")
    print(f"learned sigma = {pool[1].sigma:.4f} (objective E[delta^2 alpha] peaks at ~1.2 for beta = 2); Static move: {pool[0].sigma}")
    return simulation


if __name__ == "__main__":
    main()
