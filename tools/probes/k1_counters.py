"""Probe: K = 1 sweep with per-chain counters (packed step log) against the pool-wide counter, f64 and f32 state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.environ.get("AMC_PKG_ROOT", ROOT))
from montecarlo_amd import _capi as A
M = 10_000_000
for dtype in ("f64", "f32"):
    for counters in (False, True):
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=counters, dtype=dtype)
        e.init_uniform(-2, 2)
        t0 = time.time()
        while time.time() - t0 < 0.5:
            e.sweep(100); e.sync()
        best = 1e9
        for _ in range(4):
            e.timing_begin()
            for _ in range(600):
                e.sweep(1)
            best = min(best, e.timing_end() / 600 * 1e3)
        print(f"{dtype} per_chain_counters={counters}: {best:.2f} us per sweep (incl. amortised folds)", flush=True)
        e.close()
