#!/usr/bin/env python3
"""Ad-hoc GPU check (developer tool): HIP vs oracle bit parity on small cases + a first timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from montecarlo_amd import _capi as A

def check(name, cond):
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    return cond

ok = True
rng = np.random.default_rng(0)
# math primitives
xs = np.concatenate([rng.uniform(-708, 5, 100000), [0.0, -0.0, 1e-300, -745.0, 709.5, np.inf, -np.inf, np.nan]])
dev = A.selftest_math("exp", xs); ref = np.array([O.load().amo_exp(v) for v in xs])
ok &= check("exp bit-exact", np.array_equal(dev.view(np.uint64), ref.view(np.uint64)))
xs = np.concatenate([rng.uniform(0, 1, 100000), np.exp(rng.uniform(-700, 700, 20000)), [1.0, 2.0**-53, 5e-324, 0.0, np.inf]])
dev = A.selftest_math("log", xs); ref = np.array([O.load().amo_log(v) for v in xs])
ok &= check("log bit-exact", np.array_equal(dev.view(np.uint64), ref.view(np.uint64)))
ws = np.concatenate([(rng.integers(0, 2**53, 100000) + 1) * 2.0**-52, [0.5, 1.0, 1.5, 2.0, 2.0**-52]])
ds, dc = A.selftest_math("sinpi", ws), A.selftest_math("cospi", ws)
rs = np.array([O.sincospi(v) for v in ws])
ok &= check("sincospi bit-exact", np.array_equal(ds.view(np.uint64), rs[:, 0].copy().view(np.uint64)) and np.array_equal(dc.view(np.uint64), rs[:, 1].copy().view(np.uint64)))
a = rng.uniform(0, 80, 200000); b = rng.uniform(1e-3, 10, 200000)
ok &= check("sqrt IEEE", np.array_equal(A.selftest_math("sqrt", a), np.sqrt(a)))
ok &= check("div IEEE", np.array_equal(A.selftest_math("div", -a, b), (-a) / b))
pairs = rng.integers(0, 2**40, 1000).astype(np.uint64); ts = rng.integers(0, 2**47, 1000).astype(np.uint64)
dv = A.selftest_philox(0x123456789abcdef, pairs, ts, 5, 2)
rv = np.array([O.draw_words(0x123456789abcdef, int(p), int(t), 5, 2) for p, t in zip(pairs, ts)], dtype=np.uint32)
ok &= check("philox words", np.array_equal(dv, rv))

def parity(M, K, pot, sweeps, sweepstep, counters=True, offset=0, fused=False, beta_arr=False, seed=7):
    sigma = [0.1, 1.0, 0.3][:K]; weight = {1: [1.0], 2: [0.5, 0.5], 3: [0.2, 0.5, 0.3]}[K]
    e = A.HipEngine(n_chains=M, chain_offset=offset, n_chains_global=offset + M + 10, potential=pot, beta=2.0, sigma=sigma,
                    weight=weight, seed=seed, sweepstep=sweepstep, per_chain_counters=counters)
    o = O.OracleSim(M, chain_offset=offset, potential=pot, beta=2.0, sigma=sigma, weight=weight, seed=seed, sweepstep=sweepstep)
    e.init_uniform(-2, 2); o.init_uniform(-2, 2)
    if beta_arr:
        b = np.random.default_rng(1).uniform(1.0, 3.0, M)
        x0, _ = o.state(); e.upload_state(x0, b); o.set_beta(b)
    if fused:
        e.sweep(sweeps)
    else:
        for _ in range(sweeps): e.sweep(1)
    o.make_steps(sweeps)
    x, en = e.download_state(); xo, eo = o.state()
    good = np.array_equal(x.view(np.uint64), xo.view(np.uint64)) and np.array_equal(en.view(np.uint64), eo.view(np.uint64))
    if not good: print("  x/e mismatch", np.count_nonzero(x != xo), "of", M, np.flatnonzero(x != xo)[:10])
    acc, tot = e.counter_totals(); ao, to = o.counters()
    g2 = np.array_equal(acc, ao.sum(1)) and np.array_equal(tot, to.sum(1))
    if not g2: print("  totals mismatch", acc, ao.sum(1), tot, to.sum(1))
    good &= g2
    if counters or K > 1:
        a2, t2 = e.download_counters()
        good &= np.array_equal(a2, ao) and np.array_equal(t2, to)
    red = e.reduce()
    g3 = abs(red[0] / M - o.energy()) < 1e-10 * max(1, abs(o.energy()))
    g4 = np.allclose(red[4:] / M, o.acceptance(), rtol=1e-10, atol=0, equal_nan=True)
    if not (g3 and g4): print("  reduce mismatch", red[0] / M - o.energy(), red[4:] / M - o.acceptance())
    good &= g3 and g4
    e.close(); o.close()
    return good

for args in [(1000, 1, "harmonic", 50, 1), (1001, 1, "harmonic", 20, 3), (4097, 2, "double_well", 30, 2),
             (777, 3, "harmonic", 25, 1), (1, 1, "harmonic", 10, 1), (2, 2, "double_well", 10, 1)]:
    ok &= check(f"sweep parity M,K,pot,sweeps,sweepstep={args}", parity(*args))
ok &= check("no per-chain counters", parity(5000, 1, "harmonic", 20, 1, counters=False))
ok &= check("fused == separate", parity(3000, 2, "harmonic", 16, 2, fused=True))
ok &= check("offset shard", parity(3001, 2, "harmonic", 16, 1, offset=123456))
ok &= check("beta array", parity(2049, 1, "double_well", 16, 1, beta_arr=True))
ok &= check("large M grid-stride", parity(1_200_001, 1, "harmonic", 3, 1))

# pg estimate
M = 5001
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1, 0.4], weight=[0.5, 0.25, 0.25], seed=3)
o = O.OracleSim(M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1, 0.4], weight=[0.5, 0.25, 0.25], seed=3)
e.init_uniform(-2, 2); o.init_uniform(-2, 2); e.sweep(5); o.make_steps(5)
g = e.pg_estimate([1, 2], 3); go = o.pg_estimate([1, 2], 3)
x, _ = e.download_state(); xo, _ = o.state()
print(g, go, sep="\n")
ok &= check("pg sums", np.allclose(g, go, rtol=1e-10, atol=1e-9))
ok &= check("pg x drift bit-exact", np.array_equal(x.view(np.uint64), xo.view(np.uint64)))
e.close(); o.close()

# timing
M = 10_000_000
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2); e.sweep(20); e.sync()
for label, n, fused in [("sweepstep=1 launches", 200, False), ("fused 200", 200, True)]:
    e.timing_begin(); t0 = time.perf_counter()
    if fused: e.sweep(n)
    else:
        for _ in range(n): e.sweep(1)
    ms = e.timing_end(); wall = time.perf_counter() - t0
    print(f"{label}: {ms/n*1e3:.1f} us/sweep  {M*n/(ms*1e-3):.3e} updates/s  ({16*M*n/(ms*1e-3)/1e9:.1f} GB/s alg)  wall {wall*1e3:.1f} ms", flush=True)
red = e.reduce(); print("energy", red[0] / M, "acc", red[4] / M, "x mean", red[1] / M, "x2", red[2] / M)
e.close()
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
