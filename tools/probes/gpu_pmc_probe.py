#!/usr/bin/env python3
"""Developer tool: a short launch sequence for rocprofv3 --pmc passes (single-step and fused sweep launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A
e = A.HipEngine(n_chains=10_000_000, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2)
for _ in range(40): e.sweep(1)
e.sync()
for _ in range(3): e.sweep(40)
e.sync(); e.close()
