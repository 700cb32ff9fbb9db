"""bench.py's contract with the driver, on a real GPU: ONE JSON line with the agreed keys, for the single-process
form and for the N > 1 code path (torch.distributed over RCCL, callbacks all-reduced every 10 sweeps) forced onto one
rank -- the 8-GPU runs are the driver's, so this is the only place that path meets a device before round end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = 400_000


def run_bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "10", "--spinup-s", "0.05",
           "--chains-per-gpu", str(M), *args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def check_line(d, n_gpus, cb_every):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == 40 and d["warmup"] == 10
    assert d["unit"] == "chain-updates/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["chains_per_gpu"] == M and d["config"]["callbacks_allreduce_every"] == cb_every
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["algorithmic_bytes_per_launch"] == 16 * M
    assert rf["valu_busy"] is None or 0.3 < rf["valu_busy"] < 1.0      # from the committed PMC passes (profiles/)
    # value = chains x steps / wall; the event-timed launches cannot take longer than the wall clock around them
    assert abs(d["value"] - M * 40 / (d["ms_per_step"] * 1e-3 * 40)) < 1e-6 * d["value"]
    assert rf["avg_launch_us"] <= d["ms_per_step"] * 1e3 * 1.05
    # the chains sample exp(-2 x^2) poorly after 50 sweeps from U(-2,2), but acceptance is already at its plateau
    assert 0.90 < d["check"]["acceptance"] < 0.97 and 0.2 < d["check"]["mean_energy"] < 1.5


def test_single_process_line():
    d = run_bench({}, "--no-cpu-baseline")
    check_line(d, 1, 0)
    assert "cpu_baseline" not in d


def test_distributed_path_on_one_rank():
    env = {"AMC_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29531", "RANK": "0",
           "LOCAL_RANK": "0", "WORLD_SIZE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    d = run_bench(env, "--gpus", "1", "--no-cpu-baseline")
    check_line(d, 1, 10)
