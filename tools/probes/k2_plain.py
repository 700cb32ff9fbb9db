"""Probe: plain one-launch-per-sweep loops (K = 2 double well; fused PGMC) for a kernel-trace of a package variant
(AMC_PKG_ROOT: a copy made with tools/gpu_ab.py snapshot)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.environ.get("AMC_PKG_ROOT", ROOT))
from montecarlo_amd import _capi as A
M = 10_000_000
e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
e.init_uniform(-2, 2)
for _ in range(1500):
    e.sweep(1)
e.sync(); e.close()
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=1)
e.init_uniform(-2, 2)
e.pgmc_steps(600, [1], 1, [1], [1e-3], [0.0])
e.sync(); e.close()
