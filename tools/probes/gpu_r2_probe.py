#!/usr/bin/env python3
"""Developer tool (round 2): per-launch HIP-event timings of the three kernels VERDICT names -- K = 1 sweep (M ladder),
K = 2 sweep, fused sweep + estimator (config 5) -- over a range of AMC_BLOCKS_PER_CU."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A

def best_of(e, f, n, reps=5):
    ts = []
    for _ in range(reps):
        e.timing_begin(); f(n); ts.append(e.timing_end() / n * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]

def spin(e, s=0.4):
    t = time.perf_counter()
    while time.perf_counter() - t < s:
        e.sweep(1) if not hasattr(e, "_pg") else e.pgmc_steps(1, [1], 1, [1], [1e-3], [0.0])
        e.sync() if False else None
    e.sync()

what = sys.argv[1] if len(sys.argv) > 1 else "all"
res = {}
if what in ("all", "ladder"):
    for M in (10_000_000, 40_000_000, 160_000_000):
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
        e.init_uniform(-2, 2); spin(e)
        mn, md = best_of(e, lambda n: [e.sweep(1) for _ in range(n)], 200 if M <= 40_000_000 else 60)
        res[f"k1_M{M}"] = (mn, md)
        print(f"K=1 M={M:>11d}: min {mn:8.1f} us median {md:8.1f} us  {16*M/mn/1e3:7.1f} GB/s ({16*M/mn/1e3/8000*100:5.1f}% of 8 TB/s)", flush=True)
        e.close()
if what in ("all", "bpc", "quick"):
    M = 10_000_000
    for bpc in ((4, 5, 6, 7, 8, 10, 12) if what != "quick" else (0,)):
        if bpc: os.environ["AMC_BLOCKS_PER_CU"] = str(bpc)
        e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
        e.init_uniform(-2, 2); spin(e)
        mn, md = best_of(e, lambda n: [e.sweep(1) for _ in range(n)], 96)
        e.close()
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=1)
        e.init_uniform(-2, 2); e._pg = True; spin(e)
        mn2, md2 = best_of(e, lambda n: e.pgmc_steps(n, [1], 1, [1], [1e-3], [0.0]), 96)
        mn3, md3 = best_of(e, lambda n: [e.pg_accumulate([1], 1) for _ in range(n)], 96)
        e.close()
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
        e.init_uniform(-2, 2); spin(e)
        mn4, md4 = best_of(e, lambda n: [e.sweep(1) for _ in range(n)], 200)
        e.close()
        print(f"blocks/CU {bpc:2d}: K=2 sweep {mn:6.1f}/{md:6.1f}  pgmc fused step {mn2:6.1f}/{md2:6.1f}  estimator alone {mn3:6.1f}/{md3:6.1f}  K=1 sweep {mn4:6.1f}/{md4:6.1f} us (min/median)", flush=True)
    os.environ.pop("AMC_BLOCKS_PER_CU", None)
