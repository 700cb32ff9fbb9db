#!/usr/bin/env python3
"""Developer tool: does splitting the ensemble over S handles (S streams) hide launch ramp/tail bubbles?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A

M = int(os.environ.get("M", 10_000_000))
for S in [int(v) for v in (sys.argv[1:] or "1 2 3 4 8".split())]:
    per = (M // S) & ~1
    engs = [A.HipEngine(n_chains=per, chain_offset=i * per, n_chains_global=M, potential="harmonic", beta=2.0,
                        sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False) for i in range(S)]
    for e in engs: e.init_uniform(-2, 2)
    t0 = time.time()
    while time.time() - t0 < 0.6:
        for _ in range(100):
            for e in engs: e.sweep(1)
        for e in engs: e.sync()
    best = 1e9
    for rep in range(4):
        for e in engs: e.sync()
        t0 = time.perf_counter()
        for _ in range(1000):
            for e in engs: e.sweep(1)
        for e in engs: e.sync()
        best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
    print(f"S={S} chains/handle={per}  us per full sweep = {best:.2f}", flush=True)
    for e in engs: e.close()
