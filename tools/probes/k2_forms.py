"""Probe: the K = 2 sweep's launch forms (single step, single step with callback sums, nine steps per launch) warmed up, for a
kernel-trace of a package variant (AMC_PKG_ROOT: a copy made with tools/gpu_ab.py snapshot)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.environ.get("AMC_PKG_ROOT", ROOT))
from montecarlo_amd import _capi as A
M = 10_000_000
e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
e.init_uniform(-2, 2)
t0 = time.time()
while time.time() - t0 < 1.0:                    # clock ramp
    e.sweep(100); e.sync()
for _ in range(300):
    for _ in range(3):
        e.sweep(1)
    e.sweep(9)
    e.sweep_reduce_begin(1); e.reduce_end()
e.sync(); e.close()
