"""The engine's N > 1 code paths with TWO ranks on the one GPU of the test box.

RCCL refuses two ranks on one device and the GPU boxes have one, so nothing under `amc_comm_init` had ever met a second rank.
`tests/aux/fake_rccl.c` (test infrastructure) exports the RCCL entry points libamc.so resolves and carries all-reduce(sum)
over shared memory between the ranks' processes; `AMC_RCCL_LIBRARY` points the library at it.  What this exercises with real
kernels on a real device: shards keyed by global chain id in two processes, `sharding.connect_engine` over the launcher's
store, the estimator's in-place all-reduce on the engine's stream with n_samples of the GLOBAL ensemble, the callbacks' sums
on the communication stream between them, `amc_comm_info`, and bench.py's `--gpus 2` route end to end.  What it does not
exercise: RCCL itself and xGMI (the stand-in says version 1 so that nobody mistakes a line for the real thing).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUX = os.path.join(ROOT, "tests", "aux")


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory, gpu):
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    r = subprocess.run(["gcc", "-O2", "-shared", "-fPIC", os.path.join(AUX, "fake_rccl.c"), "-o", out, "-ldl", "-lrt", "-pthread"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


ENERGY_AFTER_BENCH = {}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(script_args, world, env, expect_ok=True):
    cmd = ["timeout", "-k", "10", "400", sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + script_args
    r = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env), capture_output=True, text=True, timeout=600)
    assert (r.returncode == 0) == expect_ok, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def test_two_ranks_share_one_gpu_pgmc_with_callbacks(fake_rccl):
    """PGMC (device-resident estimator / update over the communicator) with callbacks every 10 time steps, two ranks of 30 000
    chains each (then four of 15 000) on device 0, against the same worker on ONE rank (its own one-rank communicator): NOTHING depends on the
    sharding -- the cross-shard sums are reproducible sums (integer records merged, rounded once), so the
    learned sigma, every callback row and every chain are equal bit for bit; the communicator reports two ranks, and the
    host path (records over the same communicator, learning step on the host) gives the same bits too."""
    env = dict(AMC_TEST_GROUP="store", AMC_RCCL_LIBRARY=fake_rccl, AMC_TEST_DEVICE="0")
    worker = [os.path.join(AUX, "pgmc_comm_worker.py")]
    two = json.loads([ln for ln in launch(worker, 2, env).stdout.splitlines() if ln.startswith("{")][-1])
    one = json.loads([ln for ln in launch(worker, 1, env).stdout.splitlines() if ln.startswith("{")][-1])
    for out, world in ((two, 2), (one, 1)):
        c = out["comm"]
        assert c["world"] == world and c["connected"] and c["device_resident"] and not out["host"]["device_resident"]
        assert c["comm"]["n_ranks"] == world and c["comm"]["rank"] == 0 and c["comm"]["rccl_version"] == 1
        assert c["comm"]["librccl"] == fake_rccl
        assert c["sigma"][0] == out["host"]["sigma"][0] == 0.2 and c["sigma"][1] > 0.5
        assert c["sigma"][1] == out["host"]["sigma"][1]
        assert [v for _, v in c["energy"]] == [v for _, v in out["host"]["energy"]]
    assert two["comm"]["shard"] == [0, 30_000] and one["comm"]["shard"] == [0, 60_000]      # rank 0 reports
    # two shards against one: the same global ensemble
    assert two["comm"]["sigma"][1] == one["comm"]["sigma"][1]
    assert two["comm"]["energy"] == one["comm"]["energy"]
    assert np.array_equal(np.array([v for _, v in two["comm"]["acceptance"]]), np.array([v for _, v in one["comm"]["acceptance"]]),
                          equal_nan=True)
    # positions of the first chains (rank 0's shard starts at global chain 0 in both runs): the same bits
    assert two["comm"]["x_head"] == one["comm"]["x_head"]
    # ... and FOUR ranks of 15 000 chains each: the same sigma, the same callback rows, the same chains
    four = json.loads([ln for ln in launch(worker, 4, env).stdout.splitlines() if ln.startswith("{")][-1])
    assert four["comm"]["comm"]["n_ranks"] == 4 and four["comm"]["shard"] == [0, 15_000] and four["comm"]["connected"]
    assert four["comm"]["sigma"] == one["comm"]["sigma"] == four["host"]["sigma"]
    assert four["comm"]["energy"] == one["comm"]["energy"] and four["comm"]["x_head"] == one["comm"]["x_head"]
    assert np.array_equal(np.array([v for _, v in four["comm"]["acceptance"]]), np.array([v for _, v in one["comm"]["acceptance"]]),
                          equal_nan=True)


def test_two_ranks_two_parameter_policy(fake_rccl):
    """The same with a policy of TWO parameters and BLANPG (amc_create_vector_policy_model): one estimator launch, one gather of
    8 records and one accumulate launch per learnable move, the 2 x 2 metric inverted on the device -- two shards against one:
    the parameter vectors, the callback rows and the chains are equal bit for bit; the host path (numpy's inv) agrees to
    rounding."""
    env = dict(AMC_TEST_GROUP="store", AMC_RCCL_LIBRARY=fake_rccl, AMC_TEST_DEVICE="0", AMC_TEST_POLICY="drift2")
    worker = [os.path.join(AUX, "pgmc_comm_worker.py")]
    two = json.loads([ln for ln in launch(worker, 2, env).stdout.splitlines() if ln.startswith("{")][-1])
    one = json.loads([ln for ln in launch(worker, 1, env).stdout.splitlines() if ln.startswith("{")][-1])
    assert two["comm"]["connected"] and two["comm"]["device_resident"] and two["comm"]["comm"]["n_ranks"] == 2
    assert two["comm"]["parameters"] == one["comm"]["parameters"]
    assert two["comm"]["parameters"][0] == [float(0.0).hex(), float(0.2).hex()]                       # Static
    learned = [float.fromhex(v) for v in two["comm"]["parameters"][1]]
    assert learned != [0.05, 0.1] and 0.1 < learned[1] < 2.0
    assert two["comm"]["energy"] == one["comm"]["energy"] and two["comm"]["x_head"] == one["comm"]["x_head"]
    host = [float.fromhex(v) for v in two["host"]["parameters"][1]]
    assert np.allclose(host, learned, rtol=1e-7)          # 120 learning steps with numpy's inv against the engine's elimination
    assert two["host"]["parameters"] == one["host"]["parameters"]                                     # the host path is shard-invariant too


def test_eight_ranks_on_one_gpu(fake_rccl):
    """The machine's real shape: a communicator of EIGHT ranks.  The box lets six processes use its one GPU, so the eight ranks are
    four processes x two handles (a communicator belongs to a handle; each rank has a thread, a handle and a stream of its own:
    tests/aux/eight_rank_worker.py).  Device-resident PGMC in config 5's shape with callbacks every 10 time steps over 60 000
    chains: sigma, every callback row, the estimator's records and every chain equal ONE rank holding the whole ensemble and the
    2 x 2 arrangement, bit for bit; the communicator reports eight ranks on every rank; shards are the contiguous even-boundary
    ranges of sharding.shard_range."""
    worker = [os.path.join(AUX, "eight_rank_worker.py")]
    env = dict(AMC_RCCL_LIBRARY=fake_rccl, AMC_TEST_DEVICE="0")

    def run(processes, per_process):
        r = launch(worker, processes, dict(env, AMC_TEST_RANKS_PER_PROCESS=str(per_process)))
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    eight, four, one = run(4, 2), run(2, 2), run(1, 1)
    assert eight["n_ranks"] == 8 and eight["processes"] == 4 and len(eight["ranks"]) == 8 and four["n_ranks"] == 4 and one["n_ranks"] == 1
    for i, r in enumerate(eight["ranks"]):
        assert r["rank"] == i and r["comm"]["n_ranks"] == 8 and r["comm"]["rank"] == i and r["comm"]["librccl"] == fake_rccl
        assert r["shard"] == [i * 7500, (i + 1) * 7500]
        # replicated without a broadcast: every rank learnt the same sigma and saw the same merged callback rows
        assert r["sigma"] == eight["ranks"][0]["sigma"] and r["rows"] == eight["ranks"][0]["rows"] and r["gd"] == eight["ranks"][0]["gd"]
    ref = one["ranks"][0]
    for other in (eight, four):
        r0 = other["ranks"][0]
        assert r0["sigma"] == ref["sigma"] and r0["rows"] == ref["rows"] and r0["gd"] == ref["gd"] and r0["x_head"] == ref["x_head"]
        assert other["x_checksum"] == one["x_checksum"]                  # every chain of the ensemble, xor of the bit patterns
    assert float.fromhex(ref["sigma"][0]) == 0.2 and float.fromhex(ref["sigma"][1]) > 0.5         # Static stayed, VPG learnt
    assert len(ref["rows"]) == 12 and all(row[2] == 60_000 for row in ref["rows"])                # the callbacks saw the whole ensemble


@pytest.mark.parametrize("world", [2, 4])
def test_bench_gpus_n_over_the_stand_in(fake_rccl, world):
    """bench.py under the driver's launch line with N ranks on device 0: the RCCL route (not the store fallback), callbacks
    all-reduced every 10 sweeps, max over ranks, ONE JSON line whose config says what the communicator reported."""
    M = 400_000
    args = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "40", "--warmup", "10", "--spinup-s", "0.05", "--repeats", "2",
            "--min-gpu-seconds", "0", "--chains-per-gpu", str(M)]
    r = launch(args, world, dict(AMC_RCCL_LIBRARY=fake_rccl, AMC_BENCH_DEVICE="0", AMC_BENCH_ALLOW_FORCED_RCCL="1"))
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    c = d["config"]
    # the stand-in is named in the line, and without the test's say-so the line carries no value at all
    assert c["rccl_library_forced"] is True and c["torch_imported"] is False
    assert "/opt/rocm" in c["hip_runtime"] and c["hip_runtimes_by_rank"] == [c["hip_runtime"]]       # the stack of a single process
    if world == 2:
        r2 = launch(args, world, dict(AMC_RCCL_LIBRARY=fake_rccl, AMC_BENCH_DEVICE="0"))
        d2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.strip().startswith("{")][0])
        assert d2["value"] is None and "AMC_RCCL_LIBRARY" in d2["value_withheld"]
    assert d["n_gpus"] == world and c["chains_total"] == world * M and c["callbacks_allreduce_every"] == 10
    assert c["callbacks_allreduce_via"].startswith("rccl") and c["rccl_ranks"] == world and c["rccl_ranks_by_rank"] == [world] * world
    assert c["rccl_version"] == 1 and c["librccl"] == fake_rccl                  # the stand-in names itself
    assert 0.90 < d["check"]["acceptance"] < 0.97
    assert abs(d["value"] - world * M * 40 / (d["ms_per_step"] * 1e-3 * 40)) < 1e-6 * d["value"]
    # the ensemble is ONE ensemble whatever the number of shards: acceptance and energy of world x M chains keyed by global id
    assert d["check"]["mean_energy"] == pytest.approx(ENERGY_AFTER_BENCH.setdefault("e", d["check"]["mean_energy"]), rel=0.2)
    assert "other_configs" in d and d["other_configs"] is None                  # single-process extras stay out of N > 1 lines


def test_one_rank_cannot_join_everybody_falls_back_together(fake_rccl):
    """ncclCommInitRank fails on rank 1 only (and rank 0, like with a real communicator, waits for it in vain): after
    sharding.connect_engine NO rank holds a communicator, the sums of callbacks and estimator go over the launcher's store on
    both, and the run still equals the one-shard run -- the rank-asymmetric failure that used to leave rank 0 blocked."""
    env = dict(AMC_TEST_GROUP="store", AMC_RCCL_LIBRARY=fake_rccl, AMC_TEST_DEVICE="0", AMC_FAKE_RCCL_FAIL_RANK="1",
               AMC_FAKE_RCCL_TIMEOUT_S="3")
    worker = [os.path.join(AUX, "pgmc_comm_worker.py")]
    r = launch(worker, 2, env)
    two = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "told to fail" in r.stderr and "no RCCL communicator over the shards" in r.stderr
    for mode in ("comm", "host"):
        assert not two[mode]["connected"] and not two[mode]["device_resident"] and two[mode]["world"] == 2
        assert two[mode]["comm"] == {"n_ranks": 1, "rank": 0, "rccl_version": 0, "librccl": ""}
    one = json.loads([ln for ln in launch(worker, 1, dict(AMC_TEST_GROUP="store", AMC_RCCL_LIBRARY=fake_rccl, AMC_TEST_DEVICE="0")).stdout.splitlines()
                      if ln.startswith("{")][-1])
    assert two["comm"]["sigma"][1] == one["comm"]["sigma"][1]            # over the host path too: records, merged exactly
    assert two["comm"]["energy"] == one["comm"]["energy"]


def test_bench_withholds_its_value_when_rccl_did_not_carry_the_run(fake_rccl):
    """bench.py --gpus 2 with a rank whose ncclCommInitRank fails: every rank takes the store route together (the run completes,
    the diagnostics are all there), but the line is no measurement of the N-GPU path -- `value` is null, `value_withheld` says why,
    and the launch ends non-zero so that a driver cannot mistake it for one."""
    M = 400_000
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "10", "--spinup-s", "0.05", "--repeats", "2",
            "--min-gpu-seconds", "0", "--chains-per-gpu", str(M)]
    r = launch(args, 2, dict(AMC_RCCL_LIBRARY=fake_rccl, AMC_BENCH_DEVICE="0", AMC_BENCH_ALLOW_FORCED_RCCL="1", AMC_FAKE_RCCL_FAIL_RANK="1",
                             AMC_FAKE_RCCL_TIMEOUT_S="3"), expect_ok=False)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and "RCCL did not carry this run" in d["value_withheld"]
    assert d["config"]["rccl_ranks"] is None and "store" in d["config"]["callbacks_allreduce_via"]
    assert d["n_gpus"] == 2 and d["ms_per_step"] > 0 and 0.90 < d["check"]["acceptance"] < 0.97       # the run itself is whole
