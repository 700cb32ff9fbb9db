"""Probe: the step-log folds of a K = 2 handle for a kernel-trace of a package variant (AMC_PKG_ROOT): full 128-row folds (plain
sweeps) and 10-row ratio folds (callbacks every 10), before and after the 16-bit mark."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.environ.get("AMC_PKG_ROOT", ROOT))
from montecarlo_amd import _capi as A
M = 10_000_000
e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
e.init_uniform(-2, 2)
t0 = time.time()
while time.time() - t0 < 1.0:
    e.sweep(100); e.sync()
for late in (False, True):
    if late:
        tot = np.zeros((2, M), dtype=np.int64); tot[0] = 70_000
        e.upload_counters(np.zeros_like(tot), tot)
    for _ in range(1300):
        e.sweep(1)
    for _ in range(60):
        for _ in range(9):
            e.sweep(1)
        e.sweep_reduce_begin(1); e.reduce_end()
e.sync(); e.close()
