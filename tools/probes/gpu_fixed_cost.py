import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from montecarlo_amd import _capi as A
for M in (1000, 100_000, 1_000_000, 3_000_000, 10_000_000):
    e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
    e.init_uniform(-2, 2)
    t0 = time.time()
    while time.time() - t0 < 0.4:
        for _ in range(500): e.sweep(1)
        e.sync()
    best = 1e9
    for rep in range(3):
        e.timing_begin()
        for _ in range(2000): e.sweep(1)
        best = min(best, e.timing_end() / 2000 * 1e3)
    e.timing_begin(); e.sweep(2000); fused = e.timing_end() / 2000 * 1e3
    print(f"M={M:>9d}: {best:7.2f} us per single-sweep launch; fused {fused:7.2f} us per sweep", flush=True)
    e.close()
