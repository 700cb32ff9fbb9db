#!/usr/bin/env python3
"""Developer tool: host-side cost of the pieces of bench.py's N > 1 step loop (run under torchrun, 1 rank)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from montecarlo_amd import _capi as A, sharding
import numpy as np
e = A.HipEngine(n_chains=10_000_000, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2)
for _ in range(500): e.sweep(1)
e.sync()
def t(f, n):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("sweep(1) enqueue, empty queue  us:", t(lambda: e.sweep(1), 20)); e.sync()
print("sweep(1) enqueue, 200 deep     us:", t(lambda: e.sweep(1), 200)); e.sync()
v = np.arange(5, dtype=np.float64)
sharding.allreduce_sum(v)
print("allreduce_sum(5 doubles)       us:", t(lambda: sharding.allreduce_sum(v), 50))
def cb():
    e.sweep_reduce_begin(1); return e.reduce_end()
cb()
print("sweep_reduce_begin+reduce_end  us:", t(cb, 20), "(includes the 40 us kernel)")
e.sync(); e.close(); dist.destroy_process_group()
