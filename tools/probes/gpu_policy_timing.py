import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from montecarlo_amd import _capi as A
M = 10_000_000
for label, kw in (("standard", {}), ("scaled 0.5+x*x", dict(scale_expr="0.5 + x*x"))):
    e = A.HipEngine(n_chains=M, beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False, **kw)
    e.init_uniform(-2, 2)
    for _ in range(3000): e.sweep(1)
    e.sync()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(400): e.sweep(1)
        e.sync(); best = min(best, (time.perf_counter() - t) / 400 * 1e6)
    r = e.reduce()
    print(f"{label:16s}: {best:6.1f} us/sweep  <e>={r[0]/M:.5f} acc={r[4]/M:.4f}", flush=True)
    e.close()
