"""Probe: one launch per sweep at 1e7 chains across the engine's options -- a scan for cliffs, not a benchmark."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from montecarlo_amd import _capi as A
from montecarlo_amd.system import CustomPotential
M = 10_000_000


def run(label, steps_per_launch=1, beta_arr=False, estimator=None, **kw):
    e = A.HipEngine(n_chains=M, beta=2.0, seed=1, **kw)
    e.init_uniform(-2, 2)
    if beta_arr:
        x, _ = e.download_state()
        e.upload_state(x, np.full(M, 2.0))
    f = (lambda: e.sweep(steps_per_launch)) if estimator is None else estimator(e)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(50):
            f()
        e.sync()
    best = 1e9
    for _ in range(3):
        e.timing_begin()
        for _ in range(300):
            f()
        best = min(best, e.timing_end() / 300 * 1e3)
    print(f"{label:58s} {best / steps_per_launch:8.2f} us per MH step", flush=True)
    e.close()


one = dict(sigma=[0.1], weight=[1.0])
two = dict(sigma=[0.1, 1.0], weight=[0.5, 0.5])
run("harmonic K=1 pooled", potential="harmonic", per_chain_counters=False, **one)
run("harmonic K=1 pooled, beta array", potential="harmonic", per_chain_counters=False, beta_arr=True, **one)
run("harmonic K=1 pooled, 16 steps per launch", 16, potential="harmonic", per_chain_counters=False, **one)
run("double well K=2", potential="double_well", **two)
run("double well K=2, beta array", potential="double_well", beta_arr=True, **two)
run("double well K=2, 9 steps per launch", 9, potential="double_well", **two)
run("harmonic K=1 pooled f32", potential="harmonic", per_chain_counters=False, dtype="f32", **one)
run("double well K=2 f32", potential="double_well", dtype="f32", **two)
run("custom potential x^4-2x^2+x/4, K=1 pooled", potential=CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"), per_chain_counters=False, **one)
run("custom potential, K=2", potential=CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x"), **two)
run("harmonic K=2, scaled Gaussian policy 0.5+x^2", potential="harmonic", scale_expr="0.5 + x*x", **two)
run("harmonic K=2, estimator launch alone", potential="harmonic", estimator=lambda e: (lambda: e.pg_accumulate([1], 1)), **two)
run("harmonic K=2, fused PGMC step (VPG on move 2)", potential="harmonic", estimator=lambda e: (lambda: e.pgmc_steps(1, [1], 1, [1], [1e-3], [0.0])), **two)
run("harmonic K=2, fused PGMC step, two learnable moves", potential="harmonic", estimator=lambda e: (lambda: e.pgmc_steps(1, [0, 1], 1, [1, 1], [1e-3, 1e-3], [0.0, 0.0])), **two)
run("harmonic K=2, PGMC step, q_batch 4", potential="harmonic", estimator=lambda e: (lambda: e.pgmc_steps(1, [1], 4, [1], [1e-3], [0.0])), **two)
seven = dict(sigma=[0.2] * 7, weight=[0.4] + [0.1] * 6)
run("harmonic K=7, PGMC step, six learnable moves (pgmc_test pool)", potential="harmonic",
    estimator=lambda e: (lambda: e.pgmc_steps(1, [1, 2, 3, 4, 5, 6], 1, [1] * 6, [1e-3] * 6, [0.0] * 6)), **seven)
run("harmonic K=7, PGMC step, one learnable move", potential="harmonic", estimator=lambda e: (lambda: e.pgmc_steps(1, [1], 1, [1], [1e-3], [0.0])), **seven)
