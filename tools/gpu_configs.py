#!/usr/bin/env python3
"""Developer tool: BASELINE configs 3 and 5 at M = 1e7 through the host mirror (Simulation / run), step loop timed
with the chains left on the device.  Run under rocprofv3 --kernel-trace --stats for the per-kernel summary that is
committed as profiles/<round>_config{3,5}_kernel_stats.csv."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("AMC_PKG_ROOT", ROOT))      # A/B: a variant copy of the package (tools/gpu_ab.py snapshot)
import montecarlo_amd as ma

M = int(os.environ.get("M", 10_000_000)); steps = int(os.environ.get("STEPS", 1000))
which = sys.argv[1] if len(sys.argv) > 1 else "3"
PRECOUNT = int(os.environ.get("PRECOUNT", "0"))  # MH steps already counted per chain when the run starts (0: a fresh count)
DEFER = os.environ.get("DEFER", "1") == "1"     # StoreCallbacks writes a row when the next one is due (default) / at once


def timed(chains, al, label, pool):
    with tempfile.TemporaryDirectory() as d:
        sim = ma.Simulation(chains, al, steps, path=d)
        for alg in sim.algorithms:          # compile / allocate outside the timed region
            if PRECOUNT and hasattr(alg, "engine"):
                # start the count beyond the 16-bit mark: the regime a long run is in (high counter planes in use)
                import numpy as np
                tot = np.zeros((len(pool), M), dtype=np.int64)
                tot[0] = PRECOUNT
                alg.engine.upload_counters(np.zeros_like(tot), tot)
        t0 = time.perf_counter(); ma.run(sim); dt = time.perf_counter() - t0
        rows = open(os.path.join(d, "energy.dat")).read().strip().splitlines()
        acc = open(os.path.join(d, "acceptance.dat")).read().strip().splitlines()
    print(f"{label}{f' precount={PRECOUNT}' if PRECOUNT else ''}: M={M} steps={steps}: {dt / steps * 1e6:.1f} us/step  {M * steps / dt:.3e} chain-updates/s  "
          f"sigma={[round(m.sigma, 4) for m in pool]}  last energy row '{rows[-1]}'  acceptance '{acc[-1]}'", flush=True)


if which == "3":
    # config 3: double well U = (x^2-1)^2, beta = 2, mixed pool sigma = (0.1, 1.0), w = (0.5, 0.5), callbacks every 10
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0, potential="double_well")
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.5),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 1.0}, 0.5))
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=1, download_on_finalise=False),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
               scheduler=ma.build_schedule(steps, 100, 10), defer=DEFER))
    timed(chains, al, f"config 3 (double well, K=2, callbacks every 10, rows {'deferred' if DEFER else 'at once'})", pool)
else:
    # config 5: PGMC_harmonic_oscillator.jl:14-33 at M = 1e7: estimator + update every step, callbacks every 10
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, 0.6),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.4))
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42, download_on_finalise=False),
          dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.5)), q_batch_size=1),
          dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
               scheduler=ma.build_schedule(steps, 100, 10), defer=DEFER))
    timed(chains, al, f"config 5 (PGMC, K=2, estimator+update every step, callbacks every 10, rows {'deferred' if DEFER else 'at once'})", pool)
