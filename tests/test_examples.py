"""The two driver scripts of the reference's harmonic-oscillator example, on this engine (examples/*.py), run end to
end on the device at reduced length and checked against the reference's own known answers."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def test_mc_harmonic_oscillator_example(gpu, tmp_path, capsys):
    import mc_harmonic_oscillator as ex
    sim = ex.main(["--chains", "20000", "--steps", "3000", "--path", str(tmp_path / "mc")])
    out = capsys.readouterr().out
    assert "mean(energies), std(energies)" in out
    # the reference's density.png overlay (MC_harmonic_oscillator.jl:40-51) as a number: sampled vs sqrt(beta/pi) exp(-beta x^2)
    dev = float([ln for ln in out.splitlines() if ln.startswith("max |sampled density")][0].rsplit("=", 1)[1])
    assert dev < 0.02
    hist = sim.algorithms[2]
    assert hist.mean == pytest.approx(0.0, abs=3e-3) and hist.std == pytest.approx(0.5, abs=3e-3)   # distribution_test.jl:36-37
    rows = np.loadtxt(tmp_path / "mc" / "energy.dat", usecols=(0, 1))
    assert rows[0, 0] == 0 and rows[-1, 0] == 3000
    assert rows[rows[:, 0] >= 300, 1].mean() == pytest.approx(0.25, abs=3e-3)
    acc = open(tmp_path / "mc" / "acceptance.dat").read().splitlines()
    assert acc[0] == "0 [NaN]" and float(acc[-1].split("[")[1].strip("]")) == pytest.approx(0.93655, abs=2e-3)
    assert os.path.exists(tmp_path / "mc" / "summary.log") and os.path.exists(tmp_path / "mc" / "histogram.dat")
    # the reference script's per-chain files for the first 16 chains (MC_harmonic_oscillator.jl:24-26)
    d = tmp_path / "mc" / "trajectories" / "16"
    traj = np.loadtxt(d / "trajectory.dat")
    assert traj[0, 0] == 0 and traj[1, 0] == 300 and traj[-1, 0] == 3000 and len(traj) == len(rows)   # store_first row + schedule
    assert sorted(os.listdir(d)) == sorted(["trajectory.dat", "lastframe.dat"] + [f"restart_t{t}.dat" for t in
                                                                                 [0] + list(range(300, 3001, 300))])
    assert float(open(d / "lastframe.dat").read().split()[1]) == traj[-1, 1] == sim.chains.x[15]


def test_pgmc_harmonic_oscillator_example(gpu, tmp_path, capsys):
    import pgmc_harmonic_oscillator as ex
    sim = ex.main(["--chains", "50000", "--steps", "1500", "--eta", "0.4", "--path", str(tmp_path / "pgmc")])
    out = capsys.readouterr().out
    assert "learned sigma" in out
    pool = sim.algorithms[0].pool
    assert pool[0].sigma == 0.2 and pool[1].sigma == pytest.approx(1.2, abs=0.2)          # pgmc_test.jl:45,50
    rows = open(tmp_path / "pgmc" / "parameters" / "2" / "parameters.dat").read().splitlines()
    assert rows[0] == "0 [0.1]" and rows[-1].startswith("1500 [")


def test_mc_example_with_float32_state(gpu, tmp_path, capsys):
    """`--dtype f32`: Particle{Float32} through the same driver script (summary, trajectory rows in Float32 digits)."""
    import mc_harmonic_oscillator as ex
    sim = ex.main(["--chains", "5000", "--steps", "1500", "--dtype", "f32", "--path", str(tmp_path / "mc32")])
    capsys.readouterr()
    assert "Particle{Float32} x 5000" in open(tmp_path / "mc32" / "summary.log").read()
    assert sim.algorithms[2].std == pytest.approx(0.5, abs=1e-2)
    rows = [ln.split() for ln in open(tmp_path / "mc32" / "trajectories" / "1" / "trajectory.dat")]
    assert all(float(np.float32(float(r[1]))) == pytest.approx(float(r[1]), rel=1e-7) and len(r[1]) <= 14 for r in rows)
    assert float(np.float32(float(rows[-1][1]))) == sim.chains.x[0]


def test_pgmc_langevin_policy_example(gpu, tmp_path, capsys):
    """A policy the script defines, no derivative written by hand, in a pool with the built-in Gaussian: the step sizes are learned
    (the Langevin proposal's grows far beyond the random walk's: its drift keeps the acceptance up), the target distribution is
    kept, and the pool takes one launch per time step."""
    import pgmc_langevin_policy as ex
    sim = ex.main(["--chains", "60000", "--steps", "600", "--path", str(tmp_path / "langevin")])
    out = capsys.readouterr().out
    assert "estimator route of this pool: one launch per time step" in out and "the policy compiles" in out
    pool = sim.algorithms[0].pool
    sg, sl = float(pool[0].parameters[0]), float(pool[1].parameters[0])
    assert sg > 0.4 and sl > 0.4 and sg != sl                               # both learnt from 0.3
    rows = np.loadtxt(tmp_path / "langevin" / "energy.dat", usecols=(0, 1))
    assert rows[rows[:, 0] >= 300, 1].mean() == pytest.approx(0.25, abs=6e-3)          # <e> = 1 / (2 beta): the sampler stays exact


def test_pgmc_langevin_policy_example_with_float32_state(gpu, tmp_path, capsys):
    """The same script with Particle{Float32}: the policy's expressions see x and delta as Float32 (the generic functions of a model
    are generic in T, particle_1d.jl:9,26), the step sizes are learned, the target distribution is kept."""
    import pgmc_langevin_policy as ex
    sim = ex.main(["--chains", "40000", "--steps", "400", "--dtype", "f32", "--path", str(tmp_path / "langevin32")])
    capsys.readouterr()
    assert "Particle{Float32} x 40000" in open(tmp_path / "langevin32" / "summary.log").read()
    pool = sim.algorithms[0].pool
    assert float(pool[0].parameters[0]) > 0.4 and float(pool[1].parameters[0]) > 0.4
    x = sim.chains.x
    assert np.array_equal(x, x.astype(np.float32).astype(np.float64))
    rows = np.loadtxt(tmp_path / "langevin32" / "energy.dat", usecols=(0, 1))
    assert rows[rows[:, 0] >= 200, 1].mean() == pytest.approx(0.25, abs=8e-3)
