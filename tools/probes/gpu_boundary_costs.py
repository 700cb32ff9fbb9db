#!/usr/bin/env python3
"""Developer probe: what the host-buffer side of the C ABI costs at 1e7 chains (never part of bench.py's value):
handle creation, state upload / download over PCIe, counter download, first and cached run-time compilation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import _capi as A
from montecarlo_amd.system import CustomPotential

M = 10_000_000
def t(f, n=5):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3

t0 = time.perf_counter()
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
print(f"amc_create (first in the process, K = 1, 1e7 chains): {(time.perf_counter() - t0) * 1e3:.1f} ms")
x = np.random.default_rng(0).uniform(-2, 2, M)
print(f"amc_upload_state   80 MB host -> device: {t(lambda: e.upload_state(x)):.2f} ms")
import ctypes as C
xo, eo = np.zeros(M), np.zeros(M)                      # touched once: fresh pages would add ~5 ms of page faults per 80 MB
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
print(f"amc_download_state 80 MB x (+ 80 MB e computed on the host):  {t(lambda: e._lib.amc_download_state(e._h, dp(xo), None)):.2f} ms / {t(lambda: e._lib.amc_download_state(e._h, dp(xo), dp(eo))):.2f} ms")
e.sweep(1); e.sync()
print(f"one sweep launch + amc_sync from the host: {t(lambda: (e.sweep(1), e.sync()), 20) * 1e3:.1f} us")
e.close()
e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
e.init_uniform(-2, 2); e.sweep(10)
print(f"amc_download_counters (K = 2, 4 x 1e7 int64 out): {t(lambda: e.download_counters(), 3):.1f} ms")
e.close()
os.environ["AMC_RTC_CACHE_DIR"] = "/tmp/amc_rtc_probe"
import shutil; shutil.rmtree("/tmp/amc_rtc_probe", ignore_errors=True)
for label in ("first (hiprtc compiles)", "second handle, same process"):
    t0 = time.perf_counter()
    e = A.HipEngine(n_chains=1000, potential=CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"), beta=2.0, sigma=[0.1], weight=[1.0], seed=1)
    e.sweep(1); e.sync()
    print(f"amc_create_custom + first sweep, {label}: {(time.perf_counter() - t0) * 1e3:.0f} ms")
    e.close()
