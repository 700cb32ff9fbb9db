"""Probe: one launch per sweep at 1e7 chains for pools of 1..8 moves (incl. the amortised folds of the step log)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A
M = 10_000_000
for K in (1, 2, 4, 5, 8, 9, 12, 16, 33):
    e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1 + 0.05 * k for k in range(K)], weight=[1.0 / K] * K, seed=1,
                    per_chain_counters=True)
    e.init_uniform(-2, 2)
    t0 = time.time()
    while time.time() - t0 < 0.4:
        e.sweep(100); e.sync()
    best = 1e9
    for _ in range(3):
        e.timing_begin()
        for _ in range(512):
            e.sweep(1)
        best = min(best, e.timing_end() / 512 * 1e3)
    print(f"K={K}: {best:.2f} us per sweep incl. amortised folds", flush=True)
    e.close()
