#!/usr/bin/env python3
"""Developer tool: same-box A/B of the library variants under tools/_variants on the headline kernel at several ensemble sizes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "tools", "_variants")
CHILD = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from montecarlo_amd import _capi as A
out = []
for M in (10_000_000, 40_000_000, 160_000_000):
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
    out.append(best * 1e7 / M); e.close()
print(" ".join(f"{v:6.2f}" for v in out))
"""
names = sorted(os.listdir(VAR)); res = {n: [] for n in names}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for n in names:
        o = subprocess.run([sys.executable, "-c", CHILD, os.path.join(VAR, n)], capture_output=True, text=True)
        res[n].append(o.stdout.strip() or o.stderr.strip()[-100:])
print("us per 1e7 chains at M = 1e7, 4e7, 1.6e8; one column group per round")
for n in names: print(f"{n:12s} " + "  |  ".join(res[n]), flush=True)
