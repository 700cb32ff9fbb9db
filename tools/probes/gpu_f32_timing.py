#!/usr/bin/env python3
"""Developer tool: the sweep at M = 1e7 with Float32 state (run-time compiled kernels) beside the Float64 one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A

M = int(os.environ.get("M", 10_000_000))


def tm(e, f, n=400):
    e.sync(); t = time.perf_counter()
    for _ in range(n): f()
    e.sync(); return (time.perf_counter() - t) / n * 1e6


for dtype in ("f64", "f32"):
    for label, kw in (("K=1 pooled", dict(sigma=[0.1], weight=[1.0], per_chain_counters=False)),
                      ("K=2 double well", dict(sigma=[0.1, 1.0], weight=[0.5, 0.5], potential="double_well"))):
        t0 = time.perf_counter()
        e = A.HipEngine(n_chains=M, beta=2.0, seed=1, dtype=dtype, **kw)
        e.init_uniform(-2, 2); e.sweep(1); e.sweep(2); e.sync()
        t_build = time.perf_counter() - t0
        for _ in range(3000): e.sweep(1)          # clocks
        e.sync()
        single = min(tm(e, lambda: e.sweep(1)) for _ in range(3))
        fused = min(tm(e, lambda: e.sweep(16), 40) for _ in range(3)) / 16
        red = e.reduce()
        print(f"{dtype} {label:16s}: {single:6.1f} us/sweep ({M/single*1e6:.3e} upd/s)  fused {fused:6.1f} us/sweep  "
              f"setup {t_build:.1f} s  <e>={red[0]/M:.5f}", flush=True)
        e.close()
