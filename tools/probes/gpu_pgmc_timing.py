#!/usr/bin/env python3
"""Developer tool: BASELINE config 5 (PGMC, M = 1e7, estimator + update every sweep) timing through the host mirror."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import montecarlo_amd as ma

M = int(os.environ.get("M", 10_000_000)); steps = int(os.environ.get("STEPS", 300))
chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, 0.6),
        ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.4))
al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42, download_on_finalise=False),   # time the step loop, not a 160 MB download
      dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.5)), q_batch_size=1),
      dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
      dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=ma.build_schedule(steps, 100, 10)))
with tempfile.TemporaryDirectory() as d:
    sim = ma.Simulation(chains, al, steps, path=d)
    t0 = time.perf_counter(); ma.run(sim); dt = time.perf_counter() - t0
print(f"config 5: M={M} steps={steps}: {dt/steps*1e6:.1f} us/step  {M*steps/dt:.3e} chain-updates/s  sigma={[m.sigma for m in pool]}")
# component timings
eng = sim.algorithms[0].engine if False else None
from montecarlo_amd import _capi as A
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=1)
e.init_uniform(-2, 2); e.sweep(50); e.sync()
def tm(f, n=50):
    e.sync(); t=time.perf_counter()
    for _ in range(n): f()
    e.sync(); return (time.perf_counter()-t)/n*1e6
print("sweep(1) K=2        %.1f us" % tm(lambda: e.sweep(1)))
print("pg_estimate([1],1)  %.1f us" % tm(lambda: e.pg_estimate([1], 1)))
print("reduce()            %.1f us" % tm(lambda: e.reduce()))
print("set_parameters      %.1f us" % tm(lambda: e.set_parameters(1, [0.3])))
