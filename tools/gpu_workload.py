#!/usr/bin/env python3
"""Developer tool (rounds 2-3): the workloads profiled under rocprofv3 for profiles/r02_* / r03_*.

  ladder <M>   single-sweep launches of the headline kernel (harmonic, K = 1, pool-wide counter) at M chains
  k2           BASELINE config 3 shape: double well, K = 2 (sigma 0.1 / 1.0), one launch per sweep, callbacks every 10
  pgmc         BASELINE config 5: amc_pgmc_steps at 1e7 chains (fused sweep + estimator + update per step), callbacks every 10
  est          the estimator launch alone (amc_pg_accumulate) on the config-5 pool
  vec / vec1 / mixed   PGMC time steps of a two-parameter policy, its one-parameter script twin, a pool of two policy classes
  pgmc7        the pool of the reference's test/pgmc_test.jl:16-27 at 1e7 chains: K = 7 Gaussian displacements, six of them learnable, one
               optimiser each (VPG .. BLANPG), q_batch_size = 10 (QBATCH), sweep + estimator + learning step per time step
Environment: LAUNCHES, PIPELINED, PRECOUNT, COLS (which sums the callbacks ask for), COMM=1 (pgmc: connect a one-rank communicator).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("AMC_PKG_ROOT", ROOT))      # A/B: a variant copy of the package (tools/gpu_ab.py snapshot)
from montecarlo_amd import _capi as A

mode = sys.argv[1]
n = int(os.environ.get("LAUNCHES", "300"))
PRECOUNT = int(os.environ.get("PRECOUNT", "0"))       # MH steps already counted per chain (K = 2 workloads): 70000 = beyond the 16-bit mark
PIPELINED = os.environ.get("PIPELINED", "0") == "1"     # read a callback's sums while the next period's launches are queued


COLS = int(os.environ.get("COLS", "1"))       # the sums over x the callbacks need: 1 = sum e (callback_energy; configs 3 - 5), 7 = + the moments


def want_columns(e):
    if hasattr(e, "set_reduce_columns"):          # (A/B against libraries older than round 5: they form all three)
        e.set_reduce_columns(COLS)


def precount(e, m):
    """Start the count beyond the 16-bit mark (PRECOUNT=70000): the regime of a long run, high counter planes in use."""
    if PRECOUNT:
        import numpy as np
        tot = np.zeros((2, m), dtype=np.int64)
        tot[0] = PRECOUNT
        e.upload_counters(np.zeros_like(tot), tot)


def spin(e, seconds=0.5):
    """the GPU needs ~0.1-0.5 s of load to reach its sustained clock (same spin-up as bench.py)"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            e.sweep(1)
        e.sync()


if mode == "ladder":
    M = int(sys.argv[2])
    e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
    e.init_uniform(-2, 2)
    spin(e)
    n = max(20, min(n, int(3_000_000_000 // M)))
    for rep in range(2):
        e.timing_begin()
        for _ in range(n):
            e.sweep(1)
        us = e.timing_end() * 1e3 / n
    print(f"ladder M={M}: {us:.2f} us per launch, {16 * M / us / 1e3:.1f} GB/s")
elif mode == "k2":
    M = 10_000_000
    e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
    e.init_uniform(-2, 2)
    want_columns(e)
    precount(e, M)
    spin(e)
    for rep in range(2):
        e.timing_begin()
        pending = False
        for i in range(n):
            if (i + 1) % 10 == 0:
                if PIPELINED and pending:
                    e.reduce_end()                    # the previous callback's sums: its launches finished ten sweeps ago
                e.sweep_reduce_begin(1)
                pending = True
                if not PIPELINED:
                    e.reduce_end(); pending = False
            else:
                e.sweep(1)
        if pending:
            e.reduce_end()
        us = e.timing_end() * 1e3 / n
    print(f"k2 (config 3 shape): {us:.2f} us per time step incl. callbacks every 10{' (callback read one period later)' if PIPELINED else ''}")
elif mode in ("pgmc", "est"):
    M = 10_000_000
    e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=42)
    e.init_uniform(-2, 2)
    want_columns(e)
    precount(e, M)
    if os.environ.get("COMM", "0") == "1":                # a communicator of one rank (AMC_SHARD_ROUTE_ON_ONE_RANK=1: on the shards' route)
        e.comm_init(0, 1, A.HipEngine.comm_unique_id())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        (e.pgmc_steps(20, [1], 1, [1], [0.0], [0.0]) if mode == "pgmc" else [e.pg_accumulate([1], 1) for _ in range(20)])
        e.sync()
    for rep in range(2):
        e.timing_begin()
        if mode == "pgmc":
            pending = False
            for i in range(n // 10):
                # ten time steps; the tenth launch also forms the callback sums.  The previous callback's sums are read
                # after the first nine have been queued (the host mirror's order, policy_guided.make_steps_grouped)
                e.pgmc_steps(9, [1], 1, [1], [0.02], [0.0])
                if PIPELINED and pending:
                    e.reduce_end()                    # the previous callback's sums: queued ten time steps ago
                e.pgmc_steps(1, [1], 1, [1], [0.02], [0.0], reduce_begin=True)
                pending = True
                if not PIPELINED:
                    e.reduce_end(); pending = False
            if pending:
                e.reduce_end()
        else:
            for _ in range(n):
                e.pg_accumulate([1], 1)
        us = e.timing_end() * 1e3 / (n // 10 * 10 if mode == "pgmc" else n)
    print(f"{mode}: {us:.2f} us per {'time step incl. callbacks every 10' if mode == 'pgmc' else 'estimator launch'}{' (callback read one period later)' if PIPELINED and mode == 'pgmc' else ''}; sigma = {e.get_parameters(1)[0]:.4f}")
elif mode == "pgmc7":
    M = int(os.environ.get("CHAINS", "10000000"))
    q = int(os.environ.get("QBATCH", "10"))
    e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1] * 7, weight=[0.4] + [0.1] * 6, seed=42)
    nl = int(os.environ.get("LEARN", "6"))          # how many of the six learn (the estimator's kernel form: capacity 1, 2, 4, 8)
    learn, kinds = [1, 2, 3, 4, 5, 6][:nl], [1, 2, 3, 4, 5, 6][:nl]
    h0, h1 = [0.001, 0.001, 1e-6, 1e-2, 1e-6, 1e-6][:nl], [0.0, 0.0, 1e-6, 1e-6, 1e-6, 1e-6][:nl]
    e.init_uniform(-2, 2)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        e.pgmc_steps(2, learn, q, kinds, h0, h1)
        e.sync()
    n = min(n, 60)
    for rep in range(2):
        e.timing_begin()
        e.pgmc_steps(n, learn, q, kinds, h0, h1)
        us = e.timing_end() * 1e3 / n
    samples = M * (1 + len(learn) * q)
    print(f"pgmc7: {us:.1f} us per PGMC time step (K = 7 sweep + {len(learn)} learnable moves x {q} samples + learning steps) = "
          f"{us / (1 + len(learn) * q) / (M / 1e7):.2f} us per proposal of 1e7 chains; {samples / us * 1e6:.3e} proposals/s; sigma = {[round(float(e.get_parameters(k)[0]), 5) for k in range(7)]}")
elif mode in ("vec", "mixed", "vec1", "vec_auto", "vec1_auto", "mixed_auto", "vec_mul", "vec2"):
    # the PGMC time step of a policy with SEVERAL parameters (vec: the drift + width proposal delta = theta0 + theta1 z, one
    # learnable move, VPG), of its one-parameter twin written as a script (vec1: what the several-parameter forms are compared
    # with), and of a pool that mixes two policy classes (mixed: Gaussian + Langevin, one learnable move each)
    M = 10_000_000
    DRIFT = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
             ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"])
    GAUSS = ("sigma*z", "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0",
             "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma")
    MALA = ("-2.0*sigma*sigma*x + sigma*z",
            "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)",
            "((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(sigma*sigma*sigma) - 4.0*x*(delta + 2.0*sigma*sigma*x)/sigma - 1.0/sigma")
    # *_auto: the same policies WITHOUT their derivative expressions -- the engine differentiates logq (dual numbers, amc_dual.h)
    if mode == "vec_mul":
        # NOT the same bits: every division by a parameter-only divisor written as a multiplication by its reciprocal -- the upper
        # bound of what a division by a wave-uniform divisor can gain (profiles/NOTES_r06.md)
        DRIFT_MUL = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))*(1.0/(2.0*theta1*theta1)) - amc_log(theta1)",
                     ["(delta-theta0)*(1.0/(theta1*theta1))", "((delta-theta0)*(delta-theta0))*(1.0/(theta1*theta1*theta1)) - 1.0/theta1"])
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[[0.0, 0.5]], weight=[1.0], seed=42, proposal=DRIFT_MUL, n_params=2)
        learn, kinds, h0 = [0], [1], [1e-3]
    elif mode in ("vec", "vec_auto"):
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[[0.0, 0.5]], weight=[1.0], seed=42,
                        proposal=DRIFT if mode == "vec" else (DRIFT[0], DRIFT[1], None), n_params=2)
        learn, kinds, h0 = [0], [1], [1e-3]
    elif mode == "vec2":
        # two moves of the two-parameter policy, both learnable: one estimator launch per move (amc_pg_route says so)
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[[0.0, 0.5], [0.1, 0.9]], weight=[0.5, 0.5], seed=42, proposal=DRIFT, n_params=2)
        learn, kinds, h0 = [0, 1], [1, 1], [1e-3, 1e-3]
    elif mode in ("vec1", "vec1_auto"):
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.5], weight=[1.0], seed=42,
                        proposal=GAUSS if mode == "vec1" else (GAUSS[0], GAUSS[1], None))
        learn, kinds, h0 = [0], [1], [1e-3]
    else:
        e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.3, 0.4], weight=[0.5, 0.5], seed=42,
                        classes=[GAUSS, MALA] if mode == "mixed" else [GAUSS, (MALA[0], MALA[1], None)], class_of_move=[0, 1])
        learn, kinds, h0 = [0, 1], [1, 1], [1e-3, 1e-3]
    e.init_uniform(-2, 2)
    want_columns(e)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        e.pgmc_steps(10, learn, 1, kinds, h0, [0.0] * len(learn))
        e.sync()
    n = min(n, 400)
    for rep in range(2):
        e.timing_begin()
        e.pgmc_steps(n, learn, 1, kinds, h0, [0.0] * len(learn))
        us = e.timing_end() * 1e3 / n
    print(f"{mode}: {us:.2f} us per PGMC time step (sweep + estimator + update), parameters now {[list(e.get_parameters(k)) for k in range(e.n_moves)]}")
e.close()
