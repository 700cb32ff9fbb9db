#!/usr/bin/env python3
"""Headline benchmark: chain-updates/s of the many-chain Metropolis sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W
(for N > 1 the driver launches it under torch.distributed.run, one rank per GPU over RCCL).

A "step" is ONE make_step!(::Metropolis) (src/metropolis.jl:302-309) = one HIP launch over this rank's
shard at the reference default sweepstep = 1: every chain does one mc_step! and the state makes one
HBM round trip, so bytes/update is well defined.  Workload = BASELINE.json configs[1]:
particle_1d harmonic, beta = 2, one Gaussian displacement sigma = 0.1, M = 1e7 chains per GPU, f64,
synthetic ensemble x0 ~ U(-2, 2) generated on device (inputs resident in HBM before the timed region).
For N > 1 (configs[3]) the ensemble is N x 1e7 chains sharded by global chain id (weak scaling) and
the energy/acceptance callbacks are all-reduced over RCCL every 10 sweeps inside the timed region.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      algorithmic HBM bytes (16 B/update: read x + write x) / average launch duration measured
                with HIP events on the engine's stream over the timed region, vs 8 TB/s.
  cpu_baseline  the CPU oracle (C restatement of the reference path, kind "port": the reference is Julia,
                not runnable here) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_PER_GPU = 10_000_000
BETA, SIGMA, SEED = 2.0, 0.1, 1
BYTES_PER_UPDATE = 16            # f64 read x + write x; counters reduce in-kernel for K = 1
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
CALLBACK_EVERY_MULTI = 10        # configs[3]: callbacks all-reduced every 10 sweeps when N > 1


def cpu_allotment():
    """CPUs this process may really use: affinity mask, cut by a cgroup CPU quota when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                   # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline(budget_s: float = 10.0):
    """Oracle (checker) timed on the host: one thread (parallel=false) and OpenMP over chains (tcollect analogue) at the
    thread count that is fastest on this host -- a GPU box may show 256 logical CPUs and grant a fraction of them, so a
    short ladder of thread counts is timed first and the bounded sample then runs at the best one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.build()
    visible, quota = cpu_allotment()
    visible = min(visible, O.load().amo_max_threads())
    m = 1_000_000

    def timed(threads, budget, cap=4000):
        sim = O.OracleSim(m, potential="harmonic", beta=BETA, sigma=[SIGMA], weight=[1.0], seed=SEED)
        sim.init_uniform(-2.0, 2.0)
        sim.make_steps(1, threads)                       # warm-up / first touch
        sweeps, t0 = 0, time.perf_counter()
        while True:
            sim.make_steps(16, threads)
            sweeps += 16
            dt = time.perf_counter() - t0
            if dt >= budget or sweeps >= cap:
                break
        sim.close()
        return (m * sweeps / dt, sweeps, dt)

    out = {"single": timed(1, 3.0)}
    ladder = sorted({t for t in (8, 16, 32, 64, 128, visible, int(quota) if quota else visible) if 1 < t <= visible}
                    or {visible})
    probes = {t: timed(t, 1.5)[0] for t in ladder}
    cores = max(probes, key=probes.get)
    out["all"] = timed(cores, budget_s)
    cpu_model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    import shutil
    julia = shutil.which("julia")        # the reference itself can only be timed where Julia exists
    return {
        "value": out["all"][0], "unit": "chain-updates/s", "cores": cores, "kind": "port",
        "reference_runtime": f"julia at {julia} (reference not timed: no package depot offline)" if julia else
                             "julia not found on this host: the reference (pure Julia) cannot be timed here",
        "sample": f"oracle/amc_oracle.c (C restatement of mc_sweep!, OpenMP over chains), M=1e6 chains x "
                  f"{out['all'][1]} sweeps in {out['all'][2]:.1f} s on {cores} threads; same workload otherwise",
        "single_thread_value": out["single"][0], "cpu_model": cpu_model,
        "cpus_visible": visible, "cpu_quota": quota,
        "thread_ladder": {str(t): round(v) for t, v in probes.items()},
        "note": "SoA-free C port without the reference's per-sweep allocations: a stronger baseline than Julia",
    }


def pmc_valu_busy():
    """VALUBusy of the sweep kernel from the same committed PMC passes (which wall was hit)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("sweep_kernel_valu_busy")
    except Exception:
        return None


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), if present."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(p)).get("sweep_kernel_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--spinup-s", type=float, default=0.6,
                    help="seconds of untimed single-sweep launches before the W warm-up steps: the GPU needs "
                         "~0.1-0.5 s of load to reach its sustained clock (65 -> 56 us/sweep measured)")
    ap.add_argument("--chains-per-gpu", type=int, default=M_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # ONE JSON line on stdout: RCCL / the HIP runtime may print banners to fd 1 (e.g. RCCL's version block at
    # communicator creation), so fd 1 is pointed at stderr for the whole run and the line goes to a private copy.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    force_dist = os.environ.get("AMC_BENCH_FORCE_DIST") == "1"      # exercise the N > 1 code path on one GPU
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # The first collective builds the communicator (tens to hundreds of ms with the GPU idle).  Do it HERE: if it
        # happened in the barrier in front of the timed region, the steps would start on a clock that has fallen back
        # (measured on one rank: 31.1 instead of 29.6 us per sweep over 2000 steps).
        dist.barrier()
        torch.cuda.synchronize()

    from montecarlo_amd import _capi as A
    from montecarlo_amd import sharding

    m_local = args.chains_per_gpu
    m_global = m_local * world
    start, stop = sharding.shard_range(m_global, rank, world)
    eng = A.HipEngine(n_chains=stop - start, chain_offset=start, n_chains_global=m_global, potential="harmonic",
                      beta=BETA, sigma=[SIGMA], weight=[1.0], seed=SEED, sweepstep=1, per_chain_counters=False,
                      device=local_rank)
    eng.init_uniform(-2.0, 2.0)
    cb_every = CALLBACK_EVERY_MULTI if (world > 1 or force_dist) else 0
    if os.environ.get("AMC_BENCH_CB_EVERY"):                 # developer knob: separate the cost of the callbacks from the process group's
        cb_every = int(os.environ["AMC_BENCH_CB_EVERY"])

    pending = [False]

    def finish_callback():
        """all-reduce the callback sums enqueued one period ago (the host never drains the sweep queue)."""
        if pending[0]:
            pending[0] = False
            return sharding.allreduce_sum(eng.reduce_end())   # callback_energy + callback_acceptance, ONE all-reduce
        return None

    def step(i):
        if cb_every and (i + 1) % cb_every == 0:
            finish_callback()
            eng.sweep_reduce_begin(1)        # the sweep whose state the callbacks observe: sums formed in-kernel
            pending[0] = True
        else:
            eng.sweep(1)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()

    t_spin = time.perf_counter()                 # clock ramp (untimed), then the W warm-up steps
    while time.perf_counter() - t_spin < args.spinup_s:
        for _ in range(200):
            eng.sweep(1)
        eng.sync()
    for i in range(args.warmup):
        step(i)
    finish_callback()
    barrier()
    eng.timing_begin()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    finish_callback()                      # the last callback's all-reduce belongs to the timed region
    event_ms = eng.timing_end()            # HIP events on the engine's stream, bracketing exactly the K launches
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, event_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, event_ms = float(t[0]), float(t[1])

    red = sharding.allreduce_sum(eng.reduce())
    n = red[3]
    energy, acceptance = red[0] / n, red[4] / n

    if rank == 0:
        updates = m_global * args.steps
        launch_s = event_ms * 1e-3 / args.steps
        achieved = BYTES_PER_UPDATE * (stop - start) / launch_s / 1e9
        result = {
            "metric": "chain-updates/sec (MC sweeps x M) at M=10^7 per MI355X",
            "value": updates / elapsed,
            "unit": "chain-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "particle_1d Harmonic, beta=2.0, Gaussian Displacement sigma=0.1 (K=1), Metropolis, "
                            f"sweepstep=1, M={m_local} chains per GPU ({m_global} total), x0~U(-2,2), seed=1",
                "chains_per_gpu": m_local, "chains_total": m_global, "sweepstep": 1,
                "callbacks_allreduce_every": cb_every,
                "sharding": "contiguous global chain ids per rank; no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(), "valu_busy": pmc_valu_busy(),
                "kernel": "amc::sweep_kernel<harmonic, K=1, pool-wide counter>",
                "algorithmic_bytes_per_launch": BYTES_PER_UPDATE * (stop - start),
                "avg_launch_us": launch_s * 1e6,
                "note": "f64 VALU-bound in practice (Philox + Box-Muller + exp per update), see DESIGN.md §6",
            },
            "check": {"mean_energy": energy, "acceptance": acceptance},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
