import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("AMC_PKG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import montecarlo_amd as ma
M, steps = 10_000_000, 3000
for hist in (False, True, False, True):
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 1.0),)
    sched = ma.build_schedule(steps, 100, 10)
    al = [dict(algorithm=ma.Metropolis, pool=pool, seed=42, download_on_finalise=False),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=sched)]
    if hist:
        al.append(dict(algorithm=ma.StoreHistogram, dependencies=(ma.Metropolis,), lo=-2.0, hi=2.0, bins=200, scheduler=sched))
    with tempfile.TemporaryDirectory() as d:
        sim = ma.Simulation(chains, tuple(al), steps, path=d)
        t0 = time.perf_counter(); ma.run(sim); dt = time.perf_counter() - t0
    print(f"harmonic K=1, callbacks every 10{', histogram every 10' if hist else ''}: {dt / steps * 1e6:.1f} us per step", flush=True)
