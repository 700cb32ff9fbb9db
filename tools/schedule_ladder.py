"""Probe: config 5's pool through Simulation / run under different schedules -- a scan for cliffs in the host mirror's grouping."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("AMC_PACKAGE_ROOT", ROOT))      # another build of the package (tools/gpu_ab.py snapshot), for before / after
import montecarlo_amd as ma
M, steps = 10_000_000, 3000


def run(label, est_every, upd_every, cb_every, q=1, params_every=0):
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, 0.6),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.4))
    al = [dict(algorithm=ma.Metropolis, pool=pool, seed=42, download_on_finalise=False)]
    if est_every:
        al.append(dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=(ma.Static(), ma.VPG(0.5)), q_batch_size=q,
                       scheduler=ma.build_schedule(steps, 0, est_every)))
        al.append(dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,), scheduler=ma.build_schedule(steps, 0, upd_every)))
    if cb_every:
        al.append(dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
                       scheduler=ma.build_schedule(steps, 100, cb_every)))
    if params_every:
        al.append(dict(algorithm=ma.StoreParameters, dependencies=(ma.Metropolis,), scheduler=ma.build_schedule(steps, 100, params_every)))
    with tempfile.TemporaryDirectory() as d:
        sim = ma.Simulation(chains, tuple(al), steps, path=d)
        t0 = time.perf_counter(); ma.run(sim); dt = time.perf_counter() - t0
    print(f"{label:64s} {dt / steps * 1e6:8.1f} us per time step   sigma={[round(m.sigma, 3) for m in pool]}", flush=True)


if os.environ.get("ONLY_PARAMS"):
    run("config 5 (callbacks every 10)", 1, 1, 10)
    run("config 5 + StoreParameters every 10 (the reference's script)", 1, 1, 10, params_every=10)
    run("config 5 (callbacks every 10)", 1, 1, 10)
    run("config 5 + StoreParameters every 10 (the reference's script)", 1, 1, 10, params_every=10)
    sys.exit(0)
run("sweeps only, no callbacks", 0, 0, 0)
run("sweeps only, callbacks every 10", 0, 0, 10)
run("sweeps only, callbacks every step", 0, 0, 1)
run("estimator + update every step, callbacks every 10 (config 5)", 1, 1, 10)
run("estimator + update every step, no callbacks", 1, 1, 0)
run("estimator every step, update every 2 (pgmc_test), cb every 10", 1, 2, 10)
run("estimator + update every 2 steps, cb every 10", 2, 2, 10)
run("estimator + update every 10 steps, cb every 10", 10, 10, 10)
run("estimator + update every step, callbacks every step", 1, 1, 1)
run("estimator q_batch 4 + update every step, cb every 10", 1, 1, 10, q=4)
