import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from montecarlo_amd import _capi as A
for M in (80_000_000, 160_000_000):
    row = []
    for bpc in (6, 8, 12, 6, 8, 12, 6, 8, 12):
        os.environ["AMC_BLOCKS_PER_CU"] = str(bpc)
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
        e.init_uniform(-2, 2)
        n = max(10, min(200, int(2_000_000_000 // M)))
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for _ in range(n): e.sweep(1)
            e.sync()
        best = 1e9
        for rep in range(4):
            e.timing_begin()
            for _ in range(n): e.sweep(1)
            best = min(best, e.timing_end() * 1e3 / n)
        row.append(f"{bpc}:{best * 1e7 / M:6.2f}")
        e.close()
    print(f"M={M:>10d} us per 1e7 chains by blocks/CU  " + "  ".join(row), flush=True)
