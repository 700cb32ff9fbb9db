"""Probe: one launch per sweep across ensemble sizes -- a scan for cliffs (ns per update should fall monotonically to the 1e7 figure)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A
for M in (10, 1000, 10_000, 100_000, 300_000, 786_432, 1_000_000, 1_572_864, 2_000_000, 3_145_728, 5_000_000, 10_000_000, 20_000_000):
    out = []
    for kw in (dict(potential="harmonic", sigma=[0.1], weight=[1.0], per_chain_counters=False),
               dict(potential="double_well", sigma=[0.1, 1.0], weight=[0.5, 0.5])):
        e = A.HipEngine(n_chains=M, beta=2.0, seed=1, **kw)
        e.init_uniform(-2, 2)
        t0 = time.time()
        while time.time() - t0 < 0.25:
            for _ in range(100):
                e.sweep(1)
            e.sync()
        best = 1e9
        for _ in range(3):
            e.timing_begin()
            for _ in range(400):
                e.sweep(1)
            best = min(best, e.timing_end() / 400 * 1e3)
        out.append(best)
        e.close()
    print(f"M={M:>9d}: K=1 {out[0]:7.2f} us ({out[0] * 1e3 / M:8.3f} ns/update)   K=2 {out[1]:7.2f} us ({out[1] * 1e3 / M:8.3f} ns/update)", flush=True)
