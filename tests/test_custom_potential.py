"""User-defined potentials: `potential(x)` is a free function of the driver script in the
reference (example/particle_1d/harmonic_oscillator/MC_harmonic_oscillator.jl:4); on the GPU path it is a C
expression compiled at run time (amc_create_custom).  The oracle evaluates the SAME expression, compiled by gcc
without contraction, so chain state, energies and accept counts must agree bit for bit.

CPU part: the compile-only check, expression validation, host classes, the oracle with a custom potential against
quadrature.  GPU part (-m gpu): parity of every kernel form the custom path instantiates.
"""
import numpy as np
import pytest

from montecarlo_amd import CustomPotential, ParticleChains

TILTED = "x*x*x*x - 2.0*x*x + 0.25*x"                                 # tilted double well, + - * only
MORSE = "(1.0 - amc_exp(-(x - 0.5))) * (1.0 - amc_exp(-(x - 0.5)))"   # uses the arithmetic spec's own exp
MIXED = "amc_log(1.0 + x*x) + sqrt(fabs(x)) / (1.0 + x*x) + fma(x, x, 0.125)"
RED_RTOL = 1e-10


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# ---------------------------------------------------------------- CPU -------------------------------------------------
def test_potential_check_compiles_without_a_gpu(amc):
    for expr in (TILTED, MORSE, MIXED, "x*x"):
        assert amc.potential_check(expr) == ""                       # clean compile, empty log


@pytest.mark.parametrize("expr,needle", [
    ("x*x +", "expected expression"),          # syntax error: compiler diagnostics reach the caller
    ("x*undefined_symbol", "undeclared identifier"),
    ("y*y", "does not mention x"),
    ("2.0", "does not mention x"),
    ("x; 1", "not allowed"),
    ("x*x)\n#include <a>\n(", "not allowed"),
    ("", "1..4000"),
    ("x+" * 2500 + "x", "1..4000"),
])
def test_bad_expressions_are_rejected(amc, expr, needle):
    with pytest.raises(amc.AmcError) as ei:
        amc.potential_check(expr)
    assert needle in str(ei.value)


def test_create_custom_validates_before_touching_a_device(amc):
    with pytest.raises(amc.AmcError) as ei:
        amc.HipEngine(n_chains=8, potential=CustomPotential("no_variable_here"), sigma=[0.1], weight=[1.0])
    assert "does not mention x" in str(ei.value)
    lib = amc.load()
    cfg = amc.AmcConfig()                       # amc_create refuses the custom id without an expression
    import ctypes as C
    cfg.struct_size = C.sizeof(amc.AmcConfig)
    cfg.n_chains, cfg.n_chains_global, cfg.n_moves, cfg.sweepstep, cfg.potential = 8, 8, 1, 1, amc.AMC_POTENTIAL_CUSTOM
    s, w = (C.c_double * 1)(0.1), (C.c_double * 1)(1.0)
    cfg.sigma, cfg.weight = s, w
    h = C.c_void_p()
    assert lib.amc_create(C.byref(cfg), C.byref(h)) == -1
    assert b"amc_create_custom" in lib.amc_last_error()


def test_host_classes_accept_a_custom_potential():
    ch = ParticleChains.uniform(16, 2.0, potential=CustomPotential(TILTED))
    assert ch.potential.expr == TILTED
    with pytest.raises(ValueError):
        ParticleChains(4, 1.0, potential="not_a_potential")
    from montecarlo_amd import potential
    with pytest.raises(ValueError):
        potential(ch.potential, np.zeros(3))


def test_oracle_custom_potential_equals_builtin_and_quadrature(oracle):
    """The oracle with potential "x*x" given as an expression reproduces the built-in harmonic run bit for bit,
    and samples exp(-beta U) for the tilted well (<U>, <x> against quadrature)."""
    kw = dict(beta=2.0, sigma=[0.3], weight=[1.0], seed=11)
    a = oracle.OracleSim(500, potential=CustomPotential("x*x"), **kw)
    b = oracle.OracleSim(500, potential="harmonic", **kw)
    for s in (a, b):
        s.init_uniform(-2, 2)
        s.make_steps(50, 2)
    assert np.array_equal(bits(a.state()[0]), bits(b.state()[0])) and np.array_equal(bits(a.state()[1]), bits(b.state()[1]))
    assert np.array_equal(a.counters()[0], b.counters()[0])
    a.close(); b.close()

    beta = 2.0
    s = oracle.OracleSim(4000, potential=CustomPotential(TILTED), beta=beta, sigma=[0.4, 1.5], weight=[0.5, 0.5], seed=5)
    s.init_uniform(-2, 2)
    s.make_steps(300, 8)
    su = sx = 0.0
    n = 0
    for _ in range(40):
        s.make_steps(10, 8)
        x, e = s.state()
        su += e.sum(); sx += x.sum(); n += x.size
    xs = np.linspace(-6, 6, 200001)
    U = xs ** 4 - 2 * xs ** 2 + 0.25 * xs
    w = np.exp(-beta * (U - U.min()))
    assert abs(su / n - (U * w).sum() / w.sum()) < 0.02
    assert abs(sx / n - (xs * w).sum() / w.sum()) < 0.03
    s.close()


def test_simulation_runs_with_a_custom_potential_on_the_oracle_engine(oracle, tmp_path):
    """Host logic (Simulation / Metropolis / callbacks) with a CustomPotential, oracle as the engine (test seam)."""
    from montecarlo_amd import (Displacement, Metropolis, Move, Simulation, StandardGaussian, StoreCallbacks,
                                build_schedule, callback_energy, run)
    chains = ParticleChains.uniform(64, 2.0, potential=CustomPotential(TILTED))
    pool = [Move(Displacement(0.0), StandardGaussian(), [0.5], 1.0)]
    steps = 200
    algs = [dict(algorithm=Metropolis, pool=pool, seed=3, engine_factory=oracle.OracleEngine),
            dict(algorithm=StoreCallbacks, callbacks=(callback_energy,), scheduler=build_schedule(steps, 0, 50))]
    sim = Simulation(chains, algs, steps, path=str(tmp_path / "data"), verbose=False)
    run(sim)
    x, e = chains.x, chains.e
    assert np.array_equal(bits(e), bits(((x * x) * x) * x - (2.0 * x) * x + 0.25 * x))     # C's left-to-right order
    rows = (tmp_path / "data" / "energy.dat").read_text().strip().splitlines()
    assert len(rows) == 5 and abs(float(rows[-1].split()[1]) - e.mean()) < 1e-12


# ---------------------------------------------------------------- GPU -------------------------------------------------
def _pair(gpu, oracle, M, expr, **kw):
    pot = CustomPotential(expr)
    eng = gpu.HipEngine(n_chains=M, potential=pot, **kw)
    kw.pop("per_chain_counters", None)
    ref = oracle.OracleSim(M, potential=pot, **kw)
    return eng, ref


def _same_state(eng, ref):
    x, e = eng.download_state()
    xo, eo = ref.state()
    assert np.array_equal(bits(x), bits(xo)), "positions differ from the oracle"
    assert np.array_equal(bits(e), bits(eo)), "energies differ from the oracle"


@pytest.mark.gpu
def test_custom_x_squared_equals_builtin_harmonic(gpu):
    """The run-time compiled kernels with potential "x*x" and the offline harmonic kernels are the same program."""
    kw = dict(n_chains=20001, beta=2.0, sigma=[0.1], weight=[1.0], seed=9, per_chain_counters=False)
    a = gpu.HipEngine(potential=CustomPotential("x*x"), **kw)
    b = gpu.HipEngine(potential="harmonic", **kw)
    for e in (a, b):
        e.init_uniform(-2, 2)
        for _ in range(3):
            e.sweep(1)
        e.sweep(17)
    xa, ea = a.download_state(); xb, eb = b.download_state()
    assert np.array_equal(bits(xa), bits(xb)) and np.array_equal(bits(ea), bits(eb))
    assert np.array_equal(a.counter_totals()[0], b.counter_totals()[0])
    assert np.allclose(a.reduce(), b.reduce(), rtol=1e-13)
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("expr", [TILTED, MORSE, MIXED])
def test_custom_potential_k1_pooled_counter(gpu, oracle, expr):
    """K = 1, pool-wide counter: single-step launches, a fused launch, the in-sweep reduction."""
    M = 30011
    eng, ref = _pair(gpu, oracle, M, expr, beta=1.5, sigma=[0.35], weight=[1.0], seed=21, per_chain_counters=False)
    eng.init_uniform(-1.5, 2.5); ref.init_uniform(-1.5, 2.5)
    for _ in range(5):
        eng.sweep(1)
    eng.sweep(20)
    ref.make_steps(25, 8)
    _same_state(eng, ref)
    acc, tot = eng.counter_totals()
    ao, to = ref.counters()
    assert acc[0] == ao.sum() and tot[0] == to.sum()
    red = eng.reduce()
    assert abs(red[0] / M - ref.energy()) <= RED_RTOL * max(1.0, abs(ref.energy()))
    assert abs(red[4] / M - ref.acceptance()[0]) <= RED_RTOL
    eng.sweep_reduce_begin(1)                      # sums formed inside the sweep launch (REDUCE form)
    red2 = eng.reduce_end()
    ref.make_steps(1, 8)
    m = ref.moments()
    assert abs(red2[0] / M - ref.energy()) <= RED_RTOL * max(1.0, abs(ref.energy()))
    assert abs(red2[1] - m[0]) <= RED_RTOL * M and abs(red2[2] - m[1]) <= RED_RTOL * M
    _same_state(eng, ref)
    eng.close(); ref.close()


@pytest.mark.gpu
def test_custom_potential_mixed_pool_per_chain_beta_and_counters(gpu, oracle):
    """K = 2 (categorical pick, step log), per-chain beta, odd chain count, per-chain counters, acceptance ratios."""
    M = 12345
    rng = np.random.default_rng(3)
    beta = rng.uniform(0.5, 3.0, M)
    x0 = rng.uniform(-2, 2, M)
    eng, ref = _pair(gpu, oracle, M, TILTED, beta=1.0, sigma=[0.2, 1.1], weight=[0.7, 0.3], seed=33)
    eng.upload_state(x0, beta); ref.set_x(x0); ref.set_beta(beta)
    for _ in range(4):
        eng.sweep(1)
    eng.sweep(36)                                  # crosses a step-log fold (depth 32)
    ref.make_steps(40, 8)
    _same_state(eng, ref)
    acc, tot = eng.download_counters()
    ao, to = ref.counters()
    assert np.array_equal(acc, ao) and np.array_equal(tot, to)
    red = eng.reduce()
    assert np.allclose(red[4:] / M, ref.acceptance(), rtol=RED_RTOL, equal_nan=True)
    eng.close(); ref.close()


@pytest.mark.gpu
def test_custom_potential_k1_per_chain_counters(gpu, oracle):
    M = 7001
    eng, ref = _pair(gpu, oracle, M, MORSE, beta=2.0, sigma=[0.6], weight=[1.0], seed=8, per_chain_counters=True)
    eng.init_uniform(-1, 3); ref.init_uniform(-1, 3)
    eng.sweep(3)
    for _ in range(3):
        eng.sweep(1)
    ref.make_steps(6, 8)
    _same_state(eng, ref)
    acc, tot = eng.download_counters()
    ao, to = ref.counters()
    assert np.array_equal(acc, ao) and np.array_equal(tot, to)
    eng.close(); ref.close()


@pytest.mark.gpu
def test_custom_potential_policy_gradient_estimator(gpu, oracle):
    """pg_estimate_kernel instantiated for the custom potential (reward delta^2, state drift kept)."""
    M = 9001
    eng, ref = _pair(gpu, oracle, M, TILTED, beta=2.0, sigma=[0.3, 0.8], weight=[0.5, 0.5], seed=13)
    eng.init_uniform(-2, 2); ref.init_uniform(-2, 2)
    eng.sweep(5); ref.make_steps(5, 8)
    g, go = eng.pg_estimate([0, 1], 3), ref.pg_estimate([0, 1], 3)
    assert np.allclose(g, go, rtol=1e-11, atol=1e-11)
    _same_state(eng, ref)
    eng.close(); ref.close()


@pytest.mark.gpu
def test_custom_potential_shard_invariance_and_cache(gpu, oracle):
    """Two shards with the same expression reuse the compiled code and reproduce the one-shard run."""
    M = 10000
    kw = dict(beta=2.0, sigma=[0.4], weight=[1.0], seed=77, per_chain_counters=False)
    pot = CustomPotential(TILTED)
    whole = gpu.HipEngine(n_chains=M, potential=pot, **kw)
    parts = [gpu.HipEngine(n_chains=M // 2, chain_offset=o, n_chains_global=M, potential=pot, **kw) for o in (0, M // 2)]
    for e in [whole] + parts:
        e.init_uniform(-2, 2)
        e.sweep(1); e.sweep(9)
    xw = whole.download_state()[0]
    xp = np.concatenate([p.download_state()[0] for p in parts])
    assert np.array_equal(bits(xw), bits(xp))
    for e in [whole] + parts:
        e.close()


@pytest.mark.gpu
def test_custom_potential_compile_error_reaches_the_caller(gpu):
    with pytest.raises(gpu.AmcError) as ei:
        gpu.HipEngine(n_chains=64, potential=CustomPotential("x * (2.0 +"), sigma=[0.1], weight=[1.0])
    assert "does not compile" in str(ei.value)


@pytest.mark.gpu
def test_custom_potential_at_full_size_statistics_and_speed(gpu):
    """M = 1e7 chains in a tilted quartic well given as an expression: <U>, <x>, <x^2> against quadrature, and the
    run-time compiled sweep is as fast as the offline kernels (potential(x) is the only code that differs)."""
    import math
    from scipy import integrate
    M, beta = 10_000_000, 2.0
    U = lambda x: x ** 4 - 2.0 * x * x + 0.25 * x
    z = integrate.quad(lambda x: math.exp(-beta * U(x)), -4, 4)[0]
    mean = lambda f: integrate.quad(lambda x: f(x) * math.exp(-beta * U(x)), -4, 4)[0] / z
    e = gpu.HipEngine(n_chains=M, potential=CustomPotential(TILTED), beta=beta, sigma=[0.15, 1.2], weight=[0.5, 0.5], seed=4)
    e.init_uniform(-2, 2)
    e.sweep(600)
    r = e.reduce()
    assert r[0] / M == pytest.approx(mean(U), abs=2e-3)
    assert r[1] / M == pytest.approx(mean(lambda x: x), abs=3e-3)
    assert r[2] / M == pytest.approx(mean(lambda x: x * x), abs=3e-3)
    e.close()

    def us_per_sweep(pot):
        eng = gpu.HipEngine(n_chains=M, potential=pot, beta=beta, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
        eng.init_uniform(-2, 2)
        for _ in range(300):
            eng.sweep(1)
        eng.sync()
        best = 1e9
        for _ in range(3):
            eng.timing_begin()
            for _ in range(300):
                eng.sweep(1)
            best = min(best, eng.timing_end() / 300 * 1e3)
        eng.close()
        return best
    t_builtin, t_custom = us_per_sweep("harmonic"), us_per_sweep(CustomPotential("x*x"))
    print(f"sweep us: offline harmonic {t_builtin:.1f}, hiprtc x*x {t_custom:.1f}")
    assert t_custom < 1.10 * t_builtin


def test_disk_cache_of_compiled_potentials(amc, tmp_path, monkeypatch):
    """AMC_RTC_CACHE_DIR: a second PROCESS finds the code object on disk instead of compiling (needs no GPU)."""
    import os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from montecarlo_amd import _capi as A\n"
            "t = time.perf_counter(); A.potential_check('x*x*x + 0.125*x - 3.0'); print(time.perf_counter() - t)\n" % root)
    env = dict(os.environ, AMC_RTC_CACHE_DIR=str(tmp_path))
    times = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        times.append(float(r.stdout.strip().splitlines()[-1]))
    files = [f for f in os.listdir(tmp_path) if f.startswith("amc_rtc_") and f.endswith(".bin")]
    assert len(files) == 1 and os.path.getsize(tmp_path / files[0]) > 10_000
    assert times[1] < 0.5 * times[0]                     # a file read instead of a compile
    # a damaged entry is recompiled, not trusted
    with open(tmp_path / files[0], "r+b") as f:
        f.truncate(100)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and os.path.getsize(tmp_path / files[0]) > 10_000


@pytest.mark.gpu
def test_custom_potential_with_nan_region(gpu, oracle):
    """U(x) = x - log(x): NaN for x < 0 (the spec's log returns NaN there), +inf at 0.  A proposal into the forbidden
    region has a NaN energy, Julia's min keeps the NaN and `alpha > u` is false: rejected.  The accept filter must
    leave such decisions to the reference-ordered arithmetic; chains started at x < 0 carry NaN energies throughout."""
    M = 20011
    eng, ref = _pair(gpu, oracle, M, "x - amc_log(x)", beta=1.0, sigma=[0.8], weight=[1.0], seed=2, per_chain_counters=True)
    x0 = np.random.default_rng(1).uniform(-0.5, 3.0, M)
    x0[:3] = [0.0, -0.0, 1e-300]
    eng.upload_state(x0); ref.set_x(x0)
    for _ in range(3):
        eng.sweep(1)
    eng.sweep(17)
    ref.make_steps(20, 8)
    x, e = eng.download_state()
    xo, eo = ref.state()
    assert np.array_equal(bits(x), bits(xo))
    nan = np.isnan(eo)                                  # NaN payloads are not part of the spec: compare as NaN
    assert np.array_equal(np.isnan(e), nan) and np.array_equal(bits(e[~nan]), bits(eo[~nan]))
    acc, tot = eng.download_counters()
    ao, to = ref.counters()
    assert np.array_equal(acc, ao) and np.array_equal(tot, to)
    assert np.isnan(e[x0 < 0]).all() and np.isfinite(e[(x0 > 0.5)]).all()
    eng.close(); ref.close()


# ---------------------------------------------------------------- custom reward ---------------------------------------
REWARD = "fabs(delta) * (1.0 + 0.125 * x * x)"        # reward(action, system): |delta| weighted by the new position


def test_custom_reward_validation_needs_no_gpu(amc):
    import ctypes as C
    lib = amc.load()
    cfg = amc.AmcConfig()
    cfg.struct_size = C.sizeof(amc.AmcConfig)
    cfg.n_chains, cfg.n_chains_global, cfg.n_moves, cfg.sweepstep, cfg.potential = 8, 8, 1, 1, 0
    s, w = (C.c_double * 1)(0.1), (C.c_double * 1)(1.0)
    cfg.sigma, cfg.weight = s, w
    h = C.c_void_p()
    assert lib.amc_create_model(C.byref(cfg), None, b"x*x", C.byref(h)) == -1          # the reward must mention delta
    assert b"does not mention delta" in lib.amc_last_error()
    assert lib.amc_create_model(C.byref(cfg), b"x*x", b"delta; 1", C.byref(h)) == -1
    assert b"custom reward" in lib.amc_last_error()
    cfg.potential = 7
    assert lib.amc_create_model(C.byref(cfg), None, b"delta*delta", C.byref(h)) == -1
    assert b"names no built-in" in lib.amc_last_error()


def test_oracle_custom_reward_default_is_delta_squared(oracle):
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.5], weight=[1.0], seed=3)
    a, b = oracle.OracleSim(3000, reward_expr="delta*delta", **kw), None
    a.init_uniform(-2, 2); a.make_steps(5)
    ga = a.pg_estimate([0], 3)
    b = oracle.OracleSim(3000, **kw)                      # reward_expr None restores delta^2
    b.init_uniform(-2, 2); b.make_steps(5)
    # the same summands; the fold is a running-top sum as soon as a script-defined reward takes part and a sigma-quantum sum
    # without one: two definitions of the same sum, ~1e-12 apart at this size
    gb = b.pg_estimate([0], 3)
    assert np.allclose(ga, gb, rtol=1e-9, atol=0.0)
    assert np.array_equal(a.state()[0], b.state()[0])
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("potential", ["double_well", "custom"])
def test_custom_reward_parity_and_learning_direction(gpu, oracle, potential):
    """The estimator kernel compiled for a script-defined reward: gradient sums against the oracle (same expression by
    gcc), with a built-in potential (its own expression goes through hiprtc: states stay bit-identical to the offline
    kernels) and with a custom one."""
    M = 20011
    pot = CustomPotential(TILTED) if potential == "custom" else potential
    kw = dict(potential=pot, beta=2.0, sigma=[0.3, 0.9], weight=[0.5, 0.5], seed=17)
    eng = gpu.HipEngine(n_chains=M, reward_expr=REWARD, **kw)
    ref = oracle.OracleSim(M, reward_expr=REWARD, **kw)
    plain = gpu.HipEngine(n_chains=M, **kw) if potential != "custom" else None
    for e in (eng, ref) + ((plain,) if plain else ()):
        e.init_uniform(-2, 2)
    eng.sweep(6); ref.make_steps(6, 8)
    if plain:
        plain.sweep(6)
        assert np.array_equal(bits(plain.download_state()[0]), bits(eng.download_state()[0]))   # the sweep ignores the reward
    g, go = eng.pg_estimate([0, 1], 3), ref.pg_estimate([0, 1], 3)
    np.testing.assert_allclose(g, go, rtol=1e-11, atol=1e-11)
    _same_state(eng, ref)
    if plain:
        gp = plain.pg_estimate([0, 1], 3)
        # j differs, grad logq does not (the fold with a script-defined reward is a running-top sum, without one a
        # sigma-quantum sum: two definitions of the same sum)
        assert not np.allclose(gp[:, 0], g[:, 0]) and np.allclose(gp[:, 2], g[:, 2], rtol=1e-9, atol=0.0)
        plain.close()
    eng.close(); ref.close()
    oracle.install_custom_reward(None)
