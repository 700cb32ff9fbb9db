#!/usr/bin/env python3
"""Developer tool: sweepstep=1 launch time vs grid size (AMC_BLOCKS_PER_CU), one child process per value."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
from montecarlo_amd import _capi as A
M = int(os.environ.get("M", 10_000_000))
K = int(os.environ.get("K", 1))
sigma = [0.1, 1.0][:K]; weight = [[1.0], [0.5, 0.5]][K - 1]
e = A.HipEngine(n_chains=M, potential="harmonic" if K == 1 else "double_well", beta=2.0, sigma=sigma, weight=weight,
                seed=1, per_chain_counters=(K > 1) or os.environ.get('COUNTERS') == '1')
e.init_uniform(-2, 2)
t0 = time.time()
while time.time() - t0 < 0.6:
    for _ in range(200): e.sweep(1)
    e.sync()
best = 1e9
for rep in range(4):
    e.timing_begin()
    for _ in range(1000): e.sweep(1)
    best = min(best, e.timing_end())
print(f"{best:8.2f}")
e.close()
""" % ROOT

vals = [int(v) for v in (sys.argv[1:] or "4 5 6 7 8 10 12 14 16 21 28 32".split())]
for v in vals:
    env = dict(os.environ, AMC_BLOCKS_PER_CU=str(v))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"blocks_per_cu={v:3d}  us/sweep={out.stdout.strip()}  {out.stderr.strip()[-200:]}", flush=True)
