"""Float32 state (Particle{Float32}, particle_1d.jl:9): the HIP path against the oracle's Float32 mode, bit for bit.

The kernels are the Float64 sources compiled at run time with the state type switched (amc_kernels.h real_t); the
oracle restates Julia's promotion rules for T = Float32 with C floats (oracle/amc_oracle.c, Float32 section; pinned
against an independent numpy.float32 restatement in test_oracle_kat.py).  Host buffers stay float64 and must hold
Float32 values exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RED_RTOL = 1e-10


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def is_f32(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array_equal(bits(a), bits(a.astype(np.float32).astype(np.float64)))


def make_pair(gpu, oracle, M, **kw):
    kw.setdefault("seed", 11)
    eng = gpu.HipEngine(n_chains=M, dtype="f32", **kw)
    okw = {k: v for k, v in kw.items() if k not in ("per_chain_counters", "n_chains_global")}
    sim = oracle.OracleSim(M, dtype="f32", **okw)
    return eng, sim


@pytest.mark.parametrize("potential", ["harmonic", "double_well"])
@pytest.mark.parametrize("counters", [False, True])
def test_single_move_trajectories_bit_exact(gpu, oracle, potential, counters):
    M = 20011                                              # odd: a lone last chain
    eng, sim = make_pair(gpu, oracle, M, potential=potential, beta=2.0, sigma=[0.35], weight=[1.0],
                         per_chain_counters=counters)
    eng.init_uniform(-2.0, 2.0)
    sim.init_uniform(-2.0, 2.0)
    x, e = eng.download_state()
    xo, eo = sim.state()
    assert is_f32(x) and np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(e), bits(eo))
    for n in (1, 1, 7, 1, 64):                             # single-step launches and fused ones
        eng.sweep(n)
        sim.make_steps(n)
        x, e = eng.download_state()
        xo, eo = sim.state()
        assert is_f32(x) and is_f32(e)
        assert np.array_equal(bits(x), bits(xo))
        assert np.array_equal(bits(e), bits(eo))
    acc, tot = eng.counter_totals()
    ao, to = sim.counters()
    assert int(acc[0]) == int(ao.sum()) and int(tot[0]) == int(to.sum())
    if counters:
        a, t = eng.download_counters()
        assert np.array_equal(a, ao) and np.array_equal(t, to)
    eng.close()


def test_two_moves_per_chain_beta_and_upload_rounding(gpu, oracle):
    M = 30000
    rng = np.random.default_rng(2)
    x0 = rng.uniform(-2, 2, M)                             # not Float32 values: the upload rounds (Float32(x))
    beta = rng.uniform(0.5, 3.0, M)
    eng, sim = make_pair(gpu, oracle, M, potential="double_well", beta=1.0, sigma=[0.2, 1.1], weight=[0.3, 0.7],
                         chain_offset=1 << 33, n_chains_global=(1 << 33) + M)
    eng.upload_state(x0, beta)
    sim.set_beta(beta)
    sim.set_x(x0)
    x, _ = eng.download_state()
    assert np.array_equal(bits(x), bits(x0.astype(np.float32).astype(np.float64)))
    for n in (1, 3, 1, 1, 40):
        eng.sweep(n)
        sim.make_steps(n)
    x, e = eng.download_state()
    xo, eo = sim.state()
    assert np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(e), bits(eo))
    a, t = eng.download_counters()
    ao, to = sim.counters()
    assert np.array_equal(a, ao) and np.array_equal(t, to)
    # callback sums: Float64 accumulation of the Float32 energies on both sides
    red = eng.reduce()
    assert red[0] / M == pytest.approx(sim.energy(), rel=RED_RTOL)
    assert np.allclose(np.asarray(red[4:6]) / M, sim.acceptance(), rtol=RED_RTOL)
    mom = sim.moments()
    assert red[1] == pytest.approx(mom[0], rel=1e-9, abs=1e-6) and red[2] == pytest.approx(mom[1], rel=RED_RTOL)
    eng.close()


def test_sums_formed_in_the_sweep_launch(gpu, oracle):
    M = 50001
    eng, sim = make_pair(gpu, oracle, M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0],
                         per_chain_counters=False)
    eng.init_uniform(-2, 2)
    sim.init_uniform(-2, 2)
    eng.sweep(5)
    sim.make_steps(5)
    eng.sweep_reduce_begin(1)
    red = eng.reduce_end()
    sim.make_steps(1)
    assert red[0] / M == pytest.approx(sim.energy(), rel=RED_RTOL)
    assert red[3] == M
    assert red[4] / M == pytest.approx(float(sim.acceptance()[0]), rel=1e-12)
    eng.close()


def test_exact_accept_path_agrees_with_the_filter(gpu, monkeypatch):
    M = 40000
    kw = dict(n_chains=M, potential="double_well", beta=2.5, sigma=[0.4, 0.9], weight=[0.5, 0.5], seed=5, dtype="f32")
    a = gpu.HipEngine(**kw)
    monkeypatch.setenv("AMC_EXACT_ACCEPT", "1")
    b = gpu.HipEngine(**kw)
    monkeypatch.delenv("AMC_EXACT_ACCEPT")
    for e in (a, b):
        e.init_uniform(-2, 2)
        e.sweep(1)
        e.sweep(30)
    xa, _ = a.download_state()
    xb, _ = b.download_state()
    assert np.array_equal(bits(xa), bits(xb))
    assert np.array_equal(a.download_counters()[0], b.download_counters()[0])
    a.close()
    b.close()


def test_custom_potential_promotes_like_julia(gpu, oracle):
    """x::Float32 against Float64 literals promotes to Float64 inside the expression (C's usual arithmetic conversions
    and Julia's promotion agree); the result is converted to Float32 when it is stored in Particle.e."""
    from montecarlo_amd import CustomPotential
    M = 10001
    pot = CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x")
    eng, sim = make_pair(gpu, oracle, M, potential=pot, beta=1.5, sigma=[0.5], weight=[1.0])
    eng.init_uniform(-1.5, 1.5)
    sim.init_uniform(-1.5, 1.5)
    for n in (1, 1, 20):
        eng.sweep(n)
        sim.make_steps(n)
    x, e = eng.download_state()
    xo, eo = sim.state()
    assert is_f32(e) and np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(e), bits(eo))
    eng.close()


def test_overloaded_math_takes_its_float32_form(gpu, oracle):
    """sqrt(x), fabs(x), fma(x, x, c) with x::Float32 are Float32 operations in the kernels (C++ overloads), as they
    would be in Julia; amc_log takes and returns Float64.  The oracle compiles the same text as C++."""
    from montecarlo_amd import CustomPotential
    M = 8001
    pot = CustomPotential("sqrt(fabs(x) + 1.0f) * x * x + fma(x, x, 0.125f) - amc_log(1.0 + x*x)")
    eng, sim = make_pair(gpu, oracle, M, potential=pot, beta=1.2, sigma=[0.6], weight=[1.0])
    eng.init_uniform(-2, 2)
    sim.init_uniform(-2, 2)
    for n in (1, 12):
        eng.sweep(n)
        sim.make_steps(n)
    x, e = eng.download_state()
    xo, eo = sim.state()
    assert np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(e), bits(eo))
    eng.close()


@pytest.mark.parametrize("potential", ["harmonic", "double_well"])
def test_policy_gradient_estimator(gpu, oracle, potential):
    M = 20000
    eng, sim = make_pair(gpu, oracle, M, potential=potential, beta=2.0, sigma=[0.2, 0.6], weight=[0.6, 0.4])
    eng.init_uniform(-2, 2)
    sim.init_uniform(-2, 2)
    eng.sweep(3)
    sim.make_steps(3)
    for q in (1, 3):
        got = np.asarray(eng.pg_estimate([1, 0], q)).reshape(2, 5)
        want = sim.pg_estimate([1, 0], q)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9)
        assert np.array_equal(got[:, 4], want[:, 4])
        x, _ = eng.download_state()                        # the estimator perturbs x like the reference (always reverts)
        assert np.array_equal(bits(x), bits(sim.state()[0]))
    # one fused time step (sweep + estimator + update) equals the separate calls
    kw = dict(n_chains=M, potential=potential, beta=2.0, sigma=[0.2, 0.6], weight=[0.6, 0.4], seed=3, dtype="f32")
    a, b = gpu.HipEngine(**kw), gpu.HipEngine(**kw)
    for e in (a, b):
        e.init_uniform(-2, 2)
    a.pgmc_steps(4, [1], 1, [1], [0.3], [0.0])                      # kind 1 = VPG
    for _ in range(4):
        b.sweep(1)
        b.pg_accumulate([1], 1)
        b.pg_update([1], [1], [0.3], [0.0])
    assert np.array_equal(bits(a.download_state()[0]), bits(b.download_state()[0]))
    assert np.array_equal(a.get_parameters(1), b.get_parameters(1))
    for e in (eng, a, b):
        e.close()


def test_host_side_readers_and_checkpoint(gpu, oracle, tmp_path):
    M = 12345
    eng, sim = make_pair(gpu, oracle, M, potential="harmonic", beta=2.0, sigma=[0.3], weight=[1.0])
    eng.init_uniform(-2, 2)
    sim.init_uniform(-2, 2)
    eng.sweep(10)
    sim.make_steps(10)
    xo, _ = sim.state()
    counts = np.asarray(eng.histogram(-1.0, 1.0, 16))
    want = np.histogram(xo[(xo >= -1) & (xo < 1)], bins=16, range=(-1.0, 1.0))[0]
    assert np.array_equal(counts[:16], want)
    assert counts[16] == np.sum(xo < -1) and counts[17] == np.sum(xo >= 1)
    assert np.array_equal(bits(eng.download_strided(3, 7, 100)), bits(xo[3:3 + 700:7]))
    # download -> upload round trip is exact (Float32 values survive the Float64 host buffers)
    x, _ = eng.download_state()
    eng.upload_state(x)
    eng.sweep(5)
    sim.make_steps(5)
    assert np.array_equal(bits(eng.download_state()[0]), bits(sim.state()[0]))
    eng.close()


def test_simulation_through_the_host_mirror(gpu, oracle, tmp_path):
    """ParticleChains(dtype="f32") through Simulation / run: callbacks and final state equal the oracle's."""
    import montecarlo_amd as ma
    M, steps = 5000, 40
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0, dtype="f32")
    pool = [ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), [0.25], 1.0)]
    al = [dict(algorithm=ma.Metropolis, pool=pool, seed=9),
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
               scheduler=ma.build_schedule(steps, 0, 10))]
    sim = ma.Simulation(chains, al, steps, path=str(tmp_path))
    ma.run(sim)
    o = oracle.OracleSim(M, potential="harmonic", beta=2.0, sigma=[0.25], weight=[1.0], seed=9, dtype="f32")
    o.init_uniform(-2, 2)
    o.make_steps(steps)
    assert np.array_equal(bits(chains.x), bits(o.state()[0]))
    assert np.array_equal(bits(chains.e), bits(o.state()[1]))
    assert pool[0].accepted_calls == int(o.counters()[0].sum())
    rows = [ln.split() for ln in open(tmp_path / "energy.dat")]
    assert float(rows[-1][1]) == pytest.approx(o.energy(), rel=RED_RTOL)


def test_legacy_config_layout_is_float64(gpu):
    """A caller built against the 0.1 amc_config (88 bytes, no state_dtype) still creates Float64 handles."""
    import ctypes as C
    lib = gpu.load()
    cfg = gpu.AmcConfig()
    sig, w = (C.c_double * 1)(0.1), (C.c_double * 1)(1.0)
    cfg.struct_size = 88
    cfg.n_chains, cfg.n_chains_global, cfg.potential, cfg.n_moves = 1000, 1000, 0, 1
    cfg.beta, cfg.sigma, cfg.weight, cfg.seed, cfg.sweepstep = 2.0, sig, w, 1, 1
    cfg.state_dtype = 1                                    # beyond the declared size: must be ignored
    h = C.c_void_p()
    assert lib.amc_create(C.byref(cfg), C.byref(h)) == 0
    x = np.full(1000, 0.1)                                 # 0.1 is not a Float32 value
    out = np.empty(1000)
    dp = C.POINTER(C.c_double)
    assert lib.amc_upload_state(h, x.ctypes.data_as(dp), None) == 0
    assert lib.amc_download_state(h, out.ctypes.data_as(dp), None) == 0
    assert np.array_equal(bits(out), bits(x))
    lib.amc_destroy(h)
    cfg.struct_size = 90
    assert lib.amc_create(C.byref(cfg), C.byref(h)) != 0
