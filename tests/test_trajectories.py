"""The reference's per-chain text I/O (StoreTrajectories / StoreLastFrames / StoreBackups, src/algorithms.jl:154-303)
in its own file layout and particle_1d's row format (example/particle_1d/particle_1d.jl:63-66), on CPU through the
engine_factory seam (oracle as the engine) and on the GPU through the HIP engine."""
import os

import numpy as np
import pytest

import montecarlo_amd as ma


def build(oracle_or_none, M, steps, path, extra, seed=42, dtype="f64"):
    chains = ma.ParticleChains.uniform(M, 2.0, -2.0, 2.0, dtype=dtype)
    pool = [ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), [0.1], 1.0)]
    metro = dict(algorithm=ma.Metropolis, pool=pool, seed=seed)
    if oracle_or_none is not None:
        metro["engine_factory"] = oracle_or_none.OracleEngine
    return ma.Simulation(chains, [metro] + extra, steps, path=str(path)), chains


def check_layout(oracle, path, M, steps, seed, dtype="f64"):
    sched = list(ma.build_schedule(steps, 0, 5))
    o = oracle.OracleSim(M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=seed, dtype=dtype)
    o.init_uniform(-2, 2)
    want = {0: o.state()[0].copy()}
    for t in range(1, steps + 1):
        o.make_steps(1)
        want[t] = o.state()[0].copy()
    for c in range(M):
        d = os.path.join(path, "trajectories", str(c + 1))                      # "$c": 1-based
        rows = [ln.split() for ln in open(os.path.join(d, "trajectory.dat"))]
        ts = [int(r[0]) for r in rows]
        assert ts == [0] + [t for t in sched if t > 0]                          # store_first row at t = 0, then the schedule
        back = (lambda v: float(np.float32(float(v)))) if dtype == "f32" else float   # digits round-trip in the state's type
        for r in rows:
            assert back(r[1]) == want[int(r[0])][c]                             # shortest round-trip digits: exact
            if dtype == "f32":
                assert len(r[1]) <= 14
        last = open(os.path.join(d, "lastframe.dat")).read().split()
        assert int(last[0]) == steps and back(last[1]) == want[steps][c]
        for t in (10, 20):
            b = open(os.path.join(d, f"restart_t{t}.dat")).read().split()
            assert int(b[0]) == t and back(b[1]) == want[t][c]
        assert not os.path.exists(os.path.join(d, "restart_t0.dat"))            # store_first defaults to false (:277)


def algorithms(steps):
    return [dict(algorithm=ma.StoreTrajectories, scheduler=ma.build_schedule(steps, 0, 5)),
            dict(algorithm=ma.StoreLastFrames, scheduler=[steps]),
            dict(algorithm=ma.StoreBackups, scheduler=[10, 20])]


def test_reference_layout_and_rows_on_cpu(oracle, tmp_path):
    M, steps = 6, 30
    sim, _ = build(oracle, M, steps, tmp_path, algorithms(steps))
    ma.run(sim)
    check_layout(oracle, str(tmp_path), M, steps, 42)
    text = open(tmp_path / "summary.log").read()
    assert "StoreTrajectories" in text and "StoreLastFrames" in text and "StoreBackups" in text


def test_txt_format_prints_the_struct(oracle, tmp_path):
    M, steps = 3, 4
    sim, chains = build(oracle, M, steps, tmp_path, [dict(algorithm=ma.StoreLastFrames, fmt=ma.TXT(), scheduler=[steps])])
    ma.run(sim)
    row = open(tmp_path / "trajectories" / "2" / "lastframe.txt").read().strip()
    # generic store_trajectory (src/algorithms.jl:186-189): "$t, $system"
    assert row.startswith(f"{steps}, Particle{{Float64}}(") and row.endswith(")")
    x, beta, e = (float(v) for v in row[row.index("(") + 1:-1].split(", "))
    assert x == chains.x[1] and beta == 2.0 and e == chains.e[1]


def test_txt_format_float32_fields_print_as_literals(oracle, tmp_path):
    chains = ma.ParticleChains.uniform(2, 2.0, -2.0, 2.0, dtype="f32")
    pool = [ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), [0.1], 1.0)]
    sim = ma.Simulation(chains, [dict(algorithm=ma.Metropolis, pool=pool, seed=1, engine_factory=oracle.OracleEngine),
                                 dict(algorithm=ma.StoreLastFrames, fmt=ma.TXT(), scheduler=[3])], 3, path=str(tmp_path))
    ma.run(sim)
    row = open(tmp_path / "trajectories" / "1" / "lastframe.txt").read().strip()
    assert row.startswith("3, Particle{Float32}(") and ", 2.0f0, " in row              # show(2.0f0) inside a struct
    fields = row[row.index("(") + 1:-1].split(", ")
    assert all("f" in f for f in fields)
    assert float(np.float32(fields[0].replace("f0", "").replace("f", "e"))) == chains.x[0]


def test_a_model_can_define_its_own_row(oracle, tmp_path):
    """The reference's models override store_trajectory (particle_1d.jl:63-66 does); here that is `row=`."""
    sim, chains = build(oracle, 4, 6, tmp_path, [dict(algorithm=ma.StoreTrajectories, scheduler=[2, 4, 6], store_first=False,
                                                      row=lambda t, x, beta, e: f"{t};{x!r};{e!r};{beta}")])
    ma.run(sim)
    rows = open(tmp_path / "trajectories" / "3" / "trajectory.dat").read().splitlines()
    assert [r.split(";")[0] for r in rows] == ["2", "4", "6"] and rows[-1].endswith(";2.0")
    assert float(rows[-1].split(";")[1]) == chains.x[2] and float(rows[-1].split(";")[2]) == chains.e[2]


def test_large_ensembles_need_an_explicit_selection(oracle, tmp_path):
    sim, _ = build(oracle, 5000, 2, tmp_path, [dict(algorithm=ma.StoreTrajectories, scheduler=[1, 2])])
    with pytest.raises(ValueError, match="one file each"):
        ma.run(sim)
    sim, _ = build(oracle, 5000, 2, tmp_path / "sel", [dict(algorithm=ma.StoreTrajectories, scheduler=[1, 2],
                                                            select=(7, 1000, 5))])
    ma.run(sim)
    assert sorted(os.listdir(tmp_path / "sel" / "trajectories"), key=int) == ["8", "1008", "2008", "3008", "4008"]
    assert len(open(tmp_path / "sel" / "trajectories" / "2008" / "trajectory.dat").readlines()) == 3


def test_float32_rows_print_like_julia():
    from montecarlo_amd.trajectories import _repr_state
    assert _repr_state(float(np.float32(0.1)), "f32") == "0.1"
    assert _repr_state(float(np.float32(-1.5)), "f32") == "-1.5"
    assert _repr_state(float(np.float32(1e-5)), "f32") == "1.0f-5"
    assert _repr_state(float(np.float32(2.5e7)), "f32") == "2.5f7"
    assert _repr_state(0.0, "f32") == "0.0"
    assert _repr_state(float("nan"), "f32") == "NaN32"
    assert _repr_state(0.1, "f64") == "0.1"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_reference_layout_and_rows_on_gpu(gpu, oracle, tmp_path, dtype):
    M, steps = 10, 30
    sim, _ = build(None, M, steps, tmp_path, algorithms(steps), seed=7, dtype=dtype)
    ma.run(sim)
    check_layout(oracle, str(tmp_path), M, steps, 7, dtype=dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("beta", [2.0, 3.0])
def test_harmonic_oscillator_distribution_from_trajectory_files(gpu, tmp_path, beta):
    """test/distribution_test.jl with its own algorithm list and its own data source -- the per-chain trajectory files,
    read back like `readdlm(file)[:, 2]` -- at a fifth of its length (2e5 steps; tolerance scaled from its 1e-3 to the
    ~1.3e-3 standard error of 2e6 correlated samples: 6e-3)."""
    seed, M, steps, burn = 42, 100, 2 * 10 ** 5, 1000
    sampletimes = ma.build_schedule(steps, burn, [0, 10])
    chains = ma.ParticleChains.uniform(M, beta, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 1.0),)
    path = str(tmp_path)
    algorithm_list = (
        dict(algorithm=ma.Metropolis, pool=pool, seed=seed, parallel=False),
        dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance), scheduler=sampletimes),
        dict(algorithm=ma.StoreTrajectories, scheduler=sampletimes),
        dict(algorithm=ma.StoreBackups, scheduler=ma.build_schedule(steps, burn, steps // 10), store_first=True, store_last=True),
        dict(algorithm=ma.StoreLastFrames, scheduler=[steps]),
        dict(algorithm=ma.PrintTimeSteps, scheduler=ma.build_schedule(steps, burn, steps // 10)),
    )
    simulation = ma.Simulation(chains, algorithm_list, steps, path=path, verbose=False)
    ma.run(simulation)
    dirs = sorted(os.listdir(os.path.join(path, "trajectories")), key=int)
    assert dirs == [str(c) for c in range(1, M + 1)]
    trajectories = [np.loadtxt(os.path.join(path, "trajectories", d, "trajectory.dat"))[1:, 1] for d in dirs]   # drop the t = 0 row
    positions = np.concatenate(trajectories)
    assert positions.size == M * len(sampletimes)
    assert positions.mean() == pytest.approx(0.0, abs=6e-3)
    assert positions.std() == pytest.approx(1 / np.sqrt(2 * beta), abs=6e-3)
    assert sorted(f for f in os.listdir(os.path.join(path, "trajectories", "1")) if f.startswith("restart")) == \
        sorted(f"restart_t{t}.dat" for t in [0] + list(ma.build_schedule(steps, burn, steps // 10)))
