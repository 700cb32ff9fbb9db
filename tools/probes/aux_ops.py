"""Probe: the device-side stand-ins for per-chain text I/O at 1e7 chains (histogram, strided snapshot, counters, state)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A
M = 10_000_000
e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
e.init_uniform(-2, 2)
e.sweep(50); e.sync()


def t(label, f, n=10):
    f(); e.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    e.sync()
    print(f"{label:40s} {(time.perf_counter() - t0) / n * 1e3:9.3f} ms", flush=True)


t("histogram 200 bins", lambda: e.histogram(-2.0, 2.0, 200))
t("histogram 4096 bins", lambda: e.histogram(-2.0, 2.0, 4096))
t("download_strided 1000 chains", lambda: e.download_strided(0, M // 1000, 1000))
t("reduce (energy, acceptance, moments)", lambda: e.reduce())
t("counter_totals", lambda: e.counter_totals())
t("download_state (x, e)", lambda: e.download_state(), 3)
t("download_counters", lambda: e.download_counters(), 2)
x, _ = e.download_state()
t("upload_state", lambda: e.upload_state(x), 3)
a, tt = e.download_counters()
t("upload_counters", lambda: e.upload_counters(a, tt), 2)
e.close()
