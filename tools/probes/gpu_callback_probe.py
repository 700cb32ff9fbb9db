#!/usr/bin/env python3
"""Developer tool: bench.py's N > 1 step loop without torch (9 sweeps + 1 fused sweep/callback per period), for
rocprofv3 --kernel-trace: the per-period cost shows up as kernel durations and gaps in the trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A
e = A.HipEngine(n_chains=10_000_000, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2)
t0 = time.time()
while time.time() - t0 < 0.6:
    for _ in range(200): e.sweep(1)
    e.sync()
def run(n, cb):
    pending = False
    e.sync(); e.timing_begin()
    for i in range(n):
        if cb and (i + 1) % cb == 0:
            if pending: e.reduce_end()
            e.sweep_reduce_begin(1); pending = True
        else:
            e.sweep(1)
    if pending: e.reduce_end()
    return e.timing_end() / n * 1e3
for rep in range(2):
    print(f"plain {run(2000, 0):.2f} us/step   callbacks every 10: {run(2000, 10):.2f} us/step", flush=True)
e.close()
