"""Probe: the histogram pass alone (amc_histogram_accumulate) at 1e7 chains."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from montecarlo_amd import _capi as A
e = A.HipEngine(n_chains=10_000_000, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2); e.sweep(200); e.sync()
for _ in range(20): e.histogram_accumulate(-2.0, 2.0, 200)
e.sync()
e.timing_begin()
for _ in range(100): e.histogram_accumulate(-2.0, 2.0, 200)
print(f"{e.timing_end() * 10:.1f} us per histogram pass", flush=True)
e.close()
