import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from montecarlo_amd import _capi as A
e = A.HipEngine(n_chains=10_000_000, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2)
t0 = time.time()
while time.time() - t0 < 0.5:
    for _ in range(200): e.sweep(1)
    e.sync()
def run(mode, K=2000):
    pend = False
    e.sync(); e.timing_begin(); t0 = time.perf_counter()
    for i in range(K):
        if mode and (i + 1) % 10 == 0:
            if pend: e.reduce_end(); pend = False
            e.sweep_reduce_begin(1); pend = True
        else:
            e.sweep(1)
    if pend: e.reduce_end()
    ms = e.timing_end(); e.sync()
    return ms / K * 1e3, (time.perf_counter() - t0) / K * 1e6
for rep in range(2):
    print("plain              %.2f us (events)  %.2f us (wall)" % run(0))
    print("reduce every 10    %.2f us (events)  %.2f us (wall)" % run(1))
