"""Probe: config 5's time step with the shards' communicator in place (one rank): the three-launch form with the in-place
all-reduce on the engine's stream, against the single fused launch of an unconnected engine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from montecarlo_amd import sharding
grp = sharding.init_store_group(0, 1)
from montecarlo_amd import _capi as A
M = 10_000_000
for connected in ([bool(int(os.environ["ONLY"]))] if os.environ.get("ONLY") else (False, True)):
    e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=42)
    e.init_uniform(-2, 2)
    if connected:
        print("connect:", sharding.connect_engine(e), flush=True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
        e.pgmc_steps(20, [1], 1, [1], [0.0], [0.0]); e.sync()
    best = 1e9
    for _ in range(4):
        e.timing_begin()
        e.pgmc_steps(400, [1], 1, [1], [0.02], [0.0])
        best = min(best, e.timing_end() / 400 * 1e3)
    print(f"connected={connected}: {best:.2f} us per PGMC time step (no callbacks); sigma = {e.get_parameters(1)[0]:.4f}", flush=True)
    e.close()
