#!/usr/bin/env python3
"""Developer tool: quick parity spot-check + sweep timings (sweepstep=1 launches and fused)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from montecarlo_amd import _capi as A

def parity(M, K, pot, sweeps, counters=True):
    sigma = [0.1, 1.0][:K]; weight = [[1.0], [0.5, 0.5]][K - 1]
    e = A.HipEngine(n_chains=M, potential=pot, beta=2.0, sigma=sigma, weight=weight, seed=7, per_chain_counters=counters)
    o = O.OracleSim(M, potential=pot, beta=2.0, sigma=sigma, weight=weight, seed=7)
    e.init_uniform(-2, 2); o.init_uniform(-2, 2)
    for _ in range(sweeps): e.sweep(1)
    o.make_steps(sweeps, 4)
    x, _ = e.download_state(); xo, _ = o.state()
    acc, tot = e.counter_totals(); ao, to = o.counters()
    ok = np.array_equal(x.view(np.uint64), xo.view(np.uint64)) and np.array_equal(acc, ao.sum(1))
    e.close(); o.close()
    return ok

print("parity", parity(100001, 1, "harmonic", 5, False), parity(700001, 1, "harmonic", 3, True), parity(50001, 2, "double_well", 5), flush=True)
M = int(os.environ.get("M", 10_000_000))
for K, pot, counters in [(1, "harmonic", False), (1, "harmonic", True), (2, "double_well", True)]:
    sigma = [0.1, 1.0][:K]; weight = [[1.0], [0.5, 0.5]][K - 1]
    e = A.HipEngine(n_chains=M, potential=pot, beta=2.0, sigma=sigma, weight=weight, seed=1, per_chain_counters=counters)
    e.init_uniform(-2, 2); e.sweep(20); e.sync()
    res = []
    for label, n, fused in [("step1", 200, False), ("fused", 200, True)]:
        best = 1e9
        for rep in range(3):
            e.timing_begin()
            if fused: e.sweep(n)
            else:
                for _ in range(n): e.sweep(1)
            best = min(best, e.timing_end() / n * 1e3)
        res.append(f"{label} {best:7.1f} us  {M/best/1e3:7.2f} Gupd/s  {16*M/best/1e3/8000*100:5.1f}% of 8TB/s")
    print(f"K={K} {pot} counters={counters}: " + " | ".join(res), flush=True)
    e.close()
