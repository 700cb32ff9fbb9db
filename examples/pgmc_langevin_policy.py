#!/usr/bin/env python3
"""Policy-guided Monte Carlo with a policy the SCRIPT defines -- and no derivative written by hand.

In the reference a user defines `sample_action!` and `log_proposal_density` for a policy (example/particle_1d/particle_1d.jl:48-59
are the Gaussian displacement's) and `PolicyGradientEstimator` gets d logq / d theta from an automatic-differentiation backend
(src/PolicyGuided/gradients.jl:28-33: ForwardDiff by default).  Here the two methods are C expressions compiled for the GPU at run
time, and the derivative is likewise nobody's homework: the engine evaluates `logq` over dual numbers in the estimator kernel
(DESIGN.md section 3.11).  The policy below is a Langevin (drifted Gaussian) proposal for U(x) = x^2 at inverse temperature beta:
    delta ~ Normal(-beta * tau * 2x, sqrt(2 tau))  written with sigma = sqrt(2 tau):  delta = -beta sigma^2 x + sigma z
and its step size sigma is learned by policy gradient while the chains sample (reward: the squared displacement, as in the
reference's example).  A plain Gaussian displacement shares the pool, so the pool mixes policy types -- every learnable move of
it is taken in ONE estimator launch where that kernel form builds (Metropolis.engine.pg_route says which route a pool got).

    python examples/pgmc_langevin_policy.py [--chains 100000] [--steps 2000] [--eta 0.05]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlo_amd as ma   # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--beta", type=float, default=2.0)
    ap.add_argument("--eta", type=float, default=0.05)
    ap.add_argument("--dtype", default="f64", choices=("f64", "f32"), help="Particle{T}: Float64 (reference scripts) or Float32")
    ap.add_argument("--path", default=None)
    args = ap.parse_args(argv)
    beta, M, steps = args.beta, args.chains, args.steps

    # sample_action! and log_proposal_density of the Langevin policy (z: one standard normal variate, x: the position, sigma: the
    # parameter).  No dlogq: ScriptPolicy(sample, logq) is all the reference would ask of a user.
    langevin = ma.ScriptPolicy(sample=f"-{beta}*sigma*sigma*x + sigma*z",
                               logq=f"-((delta + {beta}*sigma*sigma*x)*(delta + {beta}*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)")
    print("the policy compiles (no GPU needed for this check):", repr(ma._capi.model_check(langevin.sample, langevin.logq)) or "clean")

    chains = ma.ParticleChains.uniform(M, beta, -2.0, 2.0, dtype=args.dtype)       # (Float32 state: x and delta are Float32 in the expressions, Julia's promotion rules)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.3}, 0.5),
            ma.Move(ma.Displacement(0.0), langevin, [0.3], 0.5))
    burn = min(200, steps // 10)
    sampletimes = ma.build_schedule(steps, burn, 10)
    path = args.path or f"data/PGMC/particle_1d/Harmonic/langevin/M{M}/seed{args.seed}"
    algorithm_list = (
        dict(algorithm=ma.Metropolis, pool=pool, seed=args.seed),
        dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.VPG(args.eta), ma.VPG(args.eta))),
        dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=sampletimes),
        dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=sampletimes),
    )
    simulation = ma.Simulation(chains, algorithm_list, steps, path=path, verbose=True)
    ma.run(simulation)

    one_launch, why = simulation.algorithms[0].engine.pg_route(2, 1, fused=True)
    print("estimator route of this pool:", "one launch per time step" if one_launch else f"one launch per move ({why[:120]})")
    rows = np.loadtxt(os.path.join(path, "energy.dat"), usecols=(0, 1))
    print(f"<e> = {rows[rows[:, 0] >= burn, 1].mean():.5f}   (target {1 / (2 * beta):.5f})")
    print("learned sigma (Gaussian displacement, Langevin):", [float(m.parameters[0]) for m in pool])
    acc = open(os.path.join(path, "acceptance.dat")).read().splitlines()[-1]
    print("acceptance at the end:", acc)
    return simulation


if __name__ == "__main__":
    main()
