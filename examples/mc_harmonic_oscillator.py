#!/usr/bin/env python3
"""Harmonic oscillator sampled by many Metropolis chains on one MI355X.

The driver the reference ships as example/particle_1d/harmonic_oscillator/MC_harmonic_oscillator.jl, on this
engine: same system (beta = 2, potential x^2), same pool (one Gaussian displacement, sigma = 0.1), same schedule
(steps = 1e5, burn 1000, callbacks in blocks [0, 10]) -- with the chain count as a parameter (the reference
uses M = 10; the engine is built for M = 1e7).  The per-chain text trajectories of the reference are replaced by
a pooled histogram (what its density plot consumes) and a strided snapshot.

    python examples/mc_harmonic_oscillator.py [--chains 10] [--steps 100000] [--path data/MC/...]
"""
import argparse
import math
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
    ap.add_argument("--path", default=None)
    ap.add_argument("--dtype", default="f64", choices=("f64", "f32"), help="Particle{T}: Float64 (reference scripts) or Float32")
    args = ap.parse_args(argv)

    seed, beta, M, steps = args.seed, args.beta, args.chains, args.steps
    burn = min(1000, steps // 10)
    chains = ma.ParticleChains.uniform(M, beta, -2.0, 2.0, dtype=args.dtype)                    # x0 = 4 rand() - 2; potential(x) = x^2
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 1.0),)
    sampletimes = ma.build_schedule(steps, burn, [0, 10])
    path = args.path or f"data/MC/particle_1d/Harmonic/beta{beta}/M{M}/seed{seed}"

    algorithm_list = (
        dict(algorithm=ma.Metropolis, pool=pool, seed=seed),
        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=sampletimes),
        dict(algorithm=ma.StoreHistogram, dependencies=(ma.Metropolis,), lo=-2.0, hi=2.0, bins=200, scheduler=sampletimes),
        dict(algorithm=ma.StoreSnapshots, dependencies=(ma.Metropolis,), stride=max(1, M // 1000),
             scheduler=ma.build_schedule(steps, burn, max(1, steps // 10))),
        dict(algorithm=ma.PrintTimeSteps, scheduler=ma.build_schedule(steps, burn, max(1, steps // 10))),
    )
    # the reference script's per-chain text files (trajectories/<c>/trajectory.dat, restart_t<t>.dat, lastframe.dat), for
    # the first chains of the ensemble: all of them at the script's own M = 10
    few = dict(select=(0, 1, min(M, 16)))
    algorithm_list += (
        dict(algorithm=ma.StoreTrajectories, scheduler=sampletimes, **few),
        dict(algorithm=ma.StoreBackups, scheduler=ma.build_schedule(steps, burn, max(1, steps // 10)), store_first=True,
             store_last=True, **few),
        dict(algorithm=ma.StoreLastFrames, scheduler=[steps], **few),
    )
    simulation = ma.Simulation(chains, algorithm_list, steps, path=path, verbose=True)
    ma.run(simulation)

    # what the reference's script reports and plots, as numbers
    rows = np.loadtxt(os.path.join(path, "energy.dat"), usecols=(0, 1))
    energies = rows[rows[:, 0] >= burn, 1]
    hist = simulation.algorithms[2]
    edges = np.linspace(-2.0, 2.0, hist.bins + 1)
    centres = 0.5 * (edges[1:] + edges[:-1])
    density = hist.global_counts[:hist.bins] / hist.global_counts[:hist.bins].sum() / (edges[1] - edges[0])
    target = np.exp(-beta * centres ** 2) * math.sqrt(beta / math.pi)
    print(f"mean(energies), std(energies) = {energies.mean():.6f}, {energies.std():.6f}   (target <e> = {1 / (2 * beta):.6f})")
    print(f"pooled x: mean {hist.mean:+.5f}, std {hist.std:.5f}   (target 0, {1 / math.sqrt(2 * beta):.5f})")
    print(f"max |sampled density - target density| over {hist.bins} bins = {np.abs(density - target).max():.4f}")
    print(f"acceptance of the move: {pool[0].accepted_calls / max(pool[0].total_calls, 1):.5f}")
    return simulation


if __name__ == "__main__":
    main()
