"""Probe: the N > 1 callback route of bench.py on one rank (communicator of one rank), block by block: where the host's time
goes (reduce_end, all-reduce, launches) and what HIP events say, with and without the event bracket of the first block."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from montecarlo_amd import sharding
grp = sharding.init_store_group(0, 1)
from montecarlo_amd import _capi as A
M = 10_000_000
eng = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
eng.init_uniform(-2, 2)
if not os.environ.get("NO_COMM"):
    print("connect:", sharding.connect_engine(eng), flush=True)
    eng.allreduce_sum([0.0])
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.6:
    for _ in range(200):
        eng.sweep(1)
    eng.sync()
pending = [False]
stat = {"red": 0.0, "ar": 0.0, "launch": 0.0}

def finish():
    if pending[0]:
        pending[0] = False
        a = time.perf_counter(); r = eng.reduce_end(); b = time.perf_counter()
        if not os.environ.get("NO_AR"): sharding.allreduce_sum(r, eng)
        c = time.perf_counter()
        stat["red"] += b - a; stat["ar"] += c - b

def block(n, events):
    for k in stat: stat[k] = 0.0
    eng.sync()
    if not os.environ.get("NO_BARRIER"): grp.barrier()
    if os.environ.get("PRE_AR"): eng.allreduce_sum([0.0])
    if events: eng.timing_begin()
    t0 = time.perf_counter()
    for i in range(n):
        if (i + 1) % 10 == 0 and not os.environ.get("NO_CB"):
            finish()
            a = time.perf_counter(); eng.sweep_reduce_begin(1); stat["launch"] += time.perf_counter() - a
            pending[0] = True
        else:
            a = time.perf_counter(); eng.sweep(1); stat["launch"] += time.perf_counter() - a
    finish()
    if events: eng.timing_mark()
    eng.sync()
    if not os.environ.get("NO_BARRIER"): grp.barrier()
    dt = time.perf_counter() - t0
    ev = eng.timing_end() if events else float("nan")
    print(f"events={int(events)} n={n}: wall {dt / n * 1e6:6.2f} us/step, events {ev * 1e3 / n:6.2f}; host per callback: reduce_end "
          f"{stat['red'] / (n / 10) * 1e6:6.1f} us, all-reduce {stat['ar'] / (n / 10) * 1e6:6.1f} us; per launch call {stat['launch'] / n * 1e6:5.2f} us", flush=True)

eng.sweep_reduce_begin(1); pending[0] = True; finish()
lens = [int(v) for v in os.environ.get("LENS", "").split(",") if v]
for j, ev in enumerate([bool(int(c)) for c in os.environ.get("SEQ", "011010")]):
    block(lens[j] if j < len(lens) else 2000, ev)
eng.close()
