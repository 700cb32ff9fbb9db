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
the energy/acceptance callbacks are all-reduced over RCCL every 10 sweeps inside the timed region -- by the engines' own
communicator (amc_comm_init / amc_allreduce_sum, on a communication stream of the engine's own), which also carries the barriers
around the timed region (one 8-byte all-reduce each; every rank stamps its own K steps after its own synchronize, the
job's time is the MAX over ranks); a plain-socket key-value store on MASTER_PORT + 1 (sharding.SocketStore, rank 0 serves it)
carries the ncclUniqueId, the set-up barriers and the max over ranks.  No torch in the worker at all: the launcher only starts
it, so libamc.so binds the system's HIP runtime and RCCL in every rank exactly as in a single process.  config.rccl_ranks is
what ncclCommCount reports for that communicator; the RCCL / HIP runtime versions and files really bound are in the line too.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      algorithmic HBM bytes (16 B/update: read x + write x) / average launch duration measured
                with HIP events on the engine's stream over the timed region, vs 8 TB/s; `regime` says which memory
                level serves the state at this size, `ladder` repeats the measurement at 4e7 and 1.6e8 chains
                (state >> the 256 MiB Infinity Cache); `traffic` / `valu_busy` come from committed rocprofv3 PMC passes
                and carry their source, commit and whether the kernel sources changed since.
  repeat        the K-step block repeated (untimed by `value`): min / median ms per step.
  fused_sweepstep16  the same ensemble with 16 MH steps per launch (reported separately, no roofline fraction).
  other_configs BASELINE configs 3 (double well, K = 2) and 5 (PGMC) end to end, us per time step with callbacks every 10.
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
    m = M_PER_GPU                    # the workload's own ensemble: 1e7 chains do not fit the host's caches either

    def timed(threads, budget, cap=4000):
        sim = O.OracleSim(m, potential="harmonic", beta=BETA, sigma=[SIGMA], weight=[1.0], seed=SEED)
        sim.init_uniform(-2.0, 2.0)
        sim.make_steps(1, threads)                       # warm-up / first touch
        sweeps, t0 = 0, time.perf_counter()
        while True:
            sim.make_steps(4, threads)
            sweeps += 4
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
    # `cores`: what the sample really had -- the fastest thread count of the ladder, but never more than the cgroup quota grants
    # (an oversubscribed ladder on a 16-CPU quota still runs on 16 CPUs' worth of time); `threads` is the thread count itself
    granted = cores if not quota else max(1, min(cores, int(quota)))
    return {
        "value": out["all"][0], "unit": "chain-updates/s", "cores": granted, "threads": cores, "kind": "port",
        "reference_runtime": f"julia at {julia} (reference not timed: no package depot offline)" if julia else
                             "julia not found on this host: the reference (pure Julia) cannot be timed here",
        "sample": f"oracle/amc_oracle.c (C restatement of mc_sweep!, OpenMP over chains), M=1e7 chains x "
                  f"{out['all'][1]} sweeps in {out['all'][2]:.1f} s on {cores} threads"
                  + (f" under a cgroup quota of {quota:g} CPUs" if quota else "") + "; same workload otherwise",
        "single_thread_value": out["single"][0], "cpu_model": cpu_model,
        "cpus_visible": visible, "cpu_quota": quota,
        "thread_ladder": {str(t): round(v) for t, v in probes.items()},
        "note": "SoA-free C port without the reference's per-sweep allocations: a stronger baseline than Julia",
    }


def kernel_source_hash():
    """Hash of the kernel sources: tells whether a committed PMC figure was taken on the kernels being timed now."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "montecarlo_amd", "csrc")
    for fn in sorted(f for f in os.listdir(csrc) if f.startswith("amc_") and f.endswith(".h") and not f.endswith(".gen.h")):      # every kernel source
        h.update(open(os.path.join(csrc, fn), "rb").read())
    return h.hexdigest()[:16]


def pmc_profile():
    """HBM bytes per launch and VALUBusy of the headline kernel from the committed rocprofv3 PMC passes (profiles/):
    bench.py cannot collect counters itself (they need the profiler around the process), so these are STATIC figures
    and say so: source file, the commit they were taken at, and whether the kernel sources have changed since."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None
    now = kernel_source_hash()
    return {"traffic": d.get("sweep_kernel_bytes_per_launch"), "valu_busy": d.get("sweep_kernel_valu_busy"),
            "source": d.get("source"), "commit": d.get("commit"), "kernel_source_hash": d.get("kernel_source_hash"),
            "kernel_sources_unchanged_since": d.get("kernel_source_hash") == now}


def regime_of(m_chains):
    state_mb = 8 * m_chains / 1e6
    return ("infinity-cache resident: %.0f MB of state < 256 MiB MALL (a launch re-reads what the previous one wrote)" % state_mb
            if 2 * state_mb < 256 * 1.048576 else
            "HBM: %.0f MB of state >> 256 MiB MALL" % state_mb)


def ladder(A, sizes, device, reps=5):
    """The same single-sweep launch at larger ensembles (HIP events; min over `reps` blocks): the fraction of the HBM
    roofline where the state no longer fits the Infinity Cache."""
    rows = []
    for m in sizes:
        try:
            e = A.HipEngine(n_chains=m, potential="harmonic", beta=BETA, sigma=[SIGMA], weight=[1.0], seed=SEED, sweepstep=1,
                            per_chain_counters=False, device=device)
        except A.AmcError as err:
            rows.append({"chains": m, "error": str(err)[:120]})
            continue
        e.init_uniform(-2.0, 2.0)
        n = max(10, min(200, int(2_000_000_000 // m)))
        e.sweep_launches(n)
        e.sync()
        ts = []
        for _ in range(reps):
            e.timing_begin()
            e.sweep_launches(n)
            ts.append(e.timing_end() * 1e3 / n)
        e.close()
        us = min(ts)
        rows.append({"chains": m, "launches_per_block": n, "us_per_launch_min": us, "us_per_launch_median": sorted(ts)[len(ts) // 2],
                     "achieved_GBps": BYTES_PER_UPDATE * m / us / 1e3, "frac": BYTES_PER_UPDATE * m / us / 1e3 / HBM_PEAK_GBS,
                     "regime": regime_of(m)})
    return rows


def widened_paths(A, m, device, reps=5):
    """The widened paths as bench lines of their own (never part of `value`): the headline's single-sweep launch with
    Float32 state (Particle{Float32}, particle_1d.jl:9,26: 8 algorithmic bytes per update) and with a script-defined
    potential compiled at run time (amc_create_custom: 16 bytes).  HIP events on the engine's stream, min over `reps` blocks."""
    out = {}
    cases = {"f32_state": (dict(potential="harmonic", dtype="f32"), 8,
                           "Particle{Float32}: x, beta, e, delta in Float32 (DESIGN.md 3.7), harmonic, K = 1, pool-wide counter"),
             "custom_potential": (dict(potential=__import__("montecarlo_amd").CustomPotential("x*x*x*x - 2.0*x*x + 0.25*x")), 16,
                                  "U(x) = x^4 - 2 x^2 + x/4 as a C expression compiled with hiprtc (amc_create_custom), K = 1, pool-wide counter")}
    for name, (kw, nbytes, what) in cases.items():
        try:
            e = A.HipEngine(n_chains=m, beta=BETA, sigma=[SIGMA], weight=[1.0], seed=SEED, sweepstep=1, per_chain_counters=False,
                            device=device, **kw)
            e.init_uniform(-2.0, 2.0)
            n = 200
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                e.sweep_launches(n)
                e.sync()
            ts = []
            for _ in range(reps):
                e.timing_begin()
                e.sweep_launches(n)
                ts.append(e.timing_end() * 1e3 / n)
            e.close()
            us = sorted(ts)[len(ts) // 2]
            out[name] = {"us_per_launch_median": us, "us_per_launch_min": min(ts), "algorithmic_bytes_per_update": nbytes,
                         "achieved_GBps": nbytes * m / us / 1e3, "frac": nbytes * m / us / 1e3 / HBM_PEAK_GBS,
                         "chain_updates_per_s": m / (us * 1e-6), "workload": what}
        except A.AmcError as err:
            out[name] = {"error": str(err)[:200]}
    return out


def other_configs(A, m, device, periods=40):
    """BASELINE configs 3 and 5 end to end at this ensemble size, next to the headline (never part of `value`): one launch per
    time step, callbacks (energy + acceptance) every 10 time steps, each callback's sums read one period late -- the form
    the host mirror's StoreCallbacks uses.  HIP events: the median of three blocks of `periods` callback periods, after at least as many (and 0.35 s) untimed.
    Two figures each: K <= 4 handles keep their per-chain counters as two u16 planes, and for the first 65 535 counted steps
    the callback's fold leaves the (all-zero) high plane alone -- 17 bytes per chain against 23 afterwards;
    `us_per_time_step` is the regime after the mark (what a long run sees), `us_per_time_step_first_65535_steps` the one before."""
    import numpy as np
    out = {}

    def run(e, period):
        e.init_uniform(-2.0, 2.0)
        e.set_reduce_columns(A.HipEngine.REDUCE_E)       # callback_energy + callback_acceptance: of the sums over x, sum e alone
        early = measure(e, period)
        # counts past the 16-bit mark: the high counter planes take part from here on (amc_upload_counters)
        tot = np.zeros((2, m), dtype=np.int64)
        tot[0] = 70_000
        e.upload_counters(np.zeros((2, m), dtype=np.int64), tot)
        del tot
        return measure(e, period), early

    def measure(e, period, spinup_s=0.35, blocks=3):
        """Median of `blocks` timed blocks of `periods` callback periods each (a block is 15-30 ms: one hiccup of a few
        milliseconds moves a single block by 10 %), after an untimed phase."""
        times = []
        for block in range(-1, blocks):
            timed = block >= 0
            pending = False
            if timed:
                e.timing_begin()
            n, t0 = 0, time.perf_counter()
            # untimed: at least `periods`, and long enough for the clock to come back up after the idle seconds of set-up
            # (allocation, the counter upload); a few thousand steps, far below the 65 535 of the first regime
            while n < periods or (not timed and time.perf_counter() - t0 < spinup_s):
                # nine time steps, then the previous callback's sums are read (one reduction in flight per engine), then the
                # tenth step, which forms the next sums: the device works through the nine while the host reads
                period(e, lambda: e.reduce_end() if pending else None)
                pending = True
                n += 1
            if pending:
                e.reduce_end()
            if timed:
                times.append(e.timing_end() * 1e3 / (10 * periods))
            else:
                e.sync()
        return sorted(times)[len(times) // 2]

    def k2_period(e, read_previous):
        e.sweep_launches(9)                                            # one launch per sweep, like the headline
        read_previous()
        e.sweep_reduce_begin(1)                                        # the tenth forms the callback sums

    def pgmc_period(e, read_previous):
        e.pgmc_steps(9, [1], 1, [1], [0.02], [0.0])                    # VPG on move 2
        read_previous()
        e.pgmc_steps(1, [1], 1, [1], [0.02], [0.0], reduce_begin=True)    # the tenth launch forms the sums

    try:
        e = A.HipEngine(n_chains=m, potential="double_well", beta=BETA, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=SEED, device=device)
        us, early = run(e, k2_period)
        out["config3_double_well_K2"] = {"us_per_time_step": us, "us_per_time_step_first_65535_steps": early, "algorithmic_bytes_per_update": 16.5,
                                         "workload": "U = (x^2-1)^2, sigma = (0.1, 1.0), w = (0.5, 0.5), per-chain counters (step log)"}
        e.close()
        e = A.HipEngine(n_chains=m, potential="harmonic", beta=BETA, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=42, device=device)
        us, early = run(e, pgmc_period)
        out["config5_pgmc"] = {"us_per_time_step": us, "us_per_time_step_first_65535_steps": early, "algorithmic_bytes_per_update": 16.5, "sigma_2_after": float(e.get_parameters(1)[0]),
                               "workload": "PGMC_harmonic_oscillator.jl pool sigma = (0.2, 0.1), w = (0.6, 0.4), optimisers (Static, VPG(0.02)), "
                                           "q_batch_size = 1; sweep + estimator + learning step in ONE launch per time step"}
        e.close()
    except A.AmcError as err:
        out["error"] = str(err)[:200]
    out.update(widened_paths(A, m, device))
    out["note"] = ("callbacks (callback_energy + callback_acceptance) every 10 time steps, each read one period late; chains x time steps "
                   "/ time = chain-updates/s of these configurations; f32_state / custom_potential: the headline's single-sweep launch on the "
                   "widened paths, achieved = their algorithmic bytes / HIP-event time per launch; not part of `value`")
    return out


def pick_device(local_rank, n_dev):
    """The device ordinal of this rank: AMC_BENCH_DEVICE if given, else LOCAL_RANK -- or 0 where the launcher hands every rank ONE
    visible device of its own (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set per rank: each process then sees a single device 0).
    None: LOCAL_RANK names a device this process cannot see."""
    if "AMC_BENCH_DEVICE" in os.environ:
        return int(os.environ["AMC_BENCH_DEVICE"])
    if 0 <= local_rank < n_dev:
        return local_rank
    if n_dev == 1 and (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")):
        return 0
    return None


def preflight(rank, local_rank, world):
    """N > 1: what this rank is about to run on, on stderr BEFORE the first collective, and the failures that can be seen
    from here as errors of their own -- so that the first run on a real 8-GPU node, if it fails, says why in its first lines
    instead of hanging in a rendezvous.  Counting devices does not initialise the GPU."""
    from montecarlo_amd import _capi as A
    n_dev = A.device_count()
    forced = os.environ.get("AMC_RCCL_LIBRARY")
    candidates = [forced] if forced else ["/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"]
    librccl = next((c for c in candidates if c and os.path.exists(c)), None)
    port = os.environ.get("MASTER_PORT")
    device = pick_device(local_rank, n_dev)
    print(f"[bench preflight rank {rank}/{world}] local_rank={local_rank} device={device} hip_devices_visible={n_dev} "
          f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')} (must be 0: dmabuf IPC) "
          f"librccl={librccl or 'NOT FOUND among ' + str(candidates)}{' (AMC_RCCL_LIBRARY)' if forced else ''} "
          f"master={os.environ.get('MASTER_ADDR', '127.0.0.1')}:{port} store_port={int(port) + 1 if port else 'MASTER_PORT unset'} "
          f"HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')} ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')}",
          file=sys.stderr, flush=True)
    if n_dev < 1:
        raise SystemExit(f"[bench rank {rank}] no HIP device visible to this process")
    if device is None or not (0 <= device < n_dev):
        raise SystemExit(f"[bench rank {rank}] device {local_rank if device is None else device} (LOCAL_RANK / AMC_BENCH_DEVICE) but {n_dev} HIP device(s) visible: "
                         f"one process per GPU needs --nproc-per-node <= the node's GPU count")
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        raise SystemExit(f"[bench rank {rank}] HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r}: RCCL between "
                         f"processes needs dmabuf IPC on this driver (export HSA_ENABLE_IPC_MODE_LEGACY=0)")
    if world > 1 and librccl is None:
        print(f"[bench preflight rank {rank}] no librccl found: the callback sums will go over the store and the line will carry no value",
              file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--spinup-s", type=float, default=0.6,
                    help="seconds of untimed single-sweep launches before the W warm-up steps: the GPU needs "
                         "~0.1-0.5 s of load to reach its sustained clock (65 -> 56 us/sweep measured)")
    ap.add_argument("--spinup-cap-s", type=float, default=3.0,
                    help="... and the ramp's upper bound: it ends earlier once three consecutive 200-launch blocks agree within 1 %%")
    ap.add_argument("--preblocks", type=int, default=12,
                    help="untimed blocks of the timed region's own shape (barrier, K steps, drain) between the ramp and the W warm-up steps")
    ap.add_argument("--chains-per-gpu", type=int, default=M_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="extra repetitions of the K-step block for min / median (not part of value)")
    ap.add_argument("--min-gpu-seconds", type=float, default=8.0,
                    help="keep repeating the K-step block (not part of value) until this much wall clock has gone by, so that a "
                         "coarse outside sampler of GPU activity sees the device busy; 0: just --repeats blocks")
    ap.add_argument("--no-ladder", action="store_true", help="skip the 4e7 / 1.6e8-chain launches behind roofline.ladder")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config-3 / config-5 end-to-end figures")
    args = ap.parse_args()

    # ONE JSON line on stdout: RCCL / the HIP runtime may print banners to fd 1 (e.g. RCCL's version block at
    # communicator creation), so fd 1 is pointed at stderr for the whole run and the line goes to a private copy.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RCCL's peer-to-peer set-up between the ranks' processes needs dmabuf IPC on this pool's driver (the legacy mode fails
    # with hipIpcGetMemHandle: invalid argument); the launcher's environment normally carries this already
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    force_dist = os.environ.get("AMC_BENCH_FORCE_DIST") == "1"      # exercise the N > 1 code path on one GPU
    from montecarlo_amd import sharding
    grp = None
    if world > 1 or force_dist:
        preflight(rank, local_rank, world)                           # before the store, before any collective: fail early and loudly
        if "MASTER_PORT" not in os.environ:                          # forced on one rank, started by hand
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        grp = sharding.init_store_group(rank, world)                 # the launcher's store: before any GPU call
        grp.barrier()

    # The CPU leg FIRST (rank 0, N = 1 only): the GPU phase then sits at the END of the run, where an outside sampler of GPU
    # activity finds it (with the baseline last, ~4 s of GPU work used to precede 22 s of host-only work).  The checker is
    # loaded and timed here and nowhere near the product path below.
    cpu_line = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_line = cpu_baseline()

    from montecarlo_amd import _capi as A
    assert world == 1 or force_dist or grp.kind == "torch" or "torch" not in sys.modules, \
        "the worker must not import torch (libamc.so binds the system's ROCm)"

    m_local = args.chains_per_gpu
    m_global = m_local * world
    start, stop = sharding.shard_range(m_global, rank, world)
    eng = A.HipEngine(n_chains=stop - start, chain_offset=start, n_chains_global=m_global, potential="harmonic",
                      beta=BETA, sigma=[SIGMA], weight=[1.0], seed=SEED, sweepstep=1, per_chain_counters=False,
                      device=(lambda d: local_rank if d is None else d)(pick_device(local_rank, A.device_count())))
    eng.init_uniform(-2.0, 2.0)
    eng.set_reduce_columns(A.HipEngine.REDUCE_E)     # the callbacks of configs 2 - 5 are callback_energy and callback_acceptance: sum e alone
    allreduce_via = "none (single process)"
    if grp is not None:
        # the shards' RCCL communicator (ncclUniqueId over the store), built HERE and used once: the first collective
        # costs tens to hundreds of ms with the GPU idle; inside the barrier in front of the timed region it would start
        # the steps on a clock that has fallen back (measured on one rank: 31.1 instead of 29.6 us per sweep)
        try:
            ok = sharding.connect_engine(eng)
            eng.allreduce_sum([0.0])
        except A.AmcError as err:
            ok = False
            print(f"[bench rank {rank}] RCCL communicator not available ({err}); callback sums go over the launcher's store",
                  file=sys.stderr)
            eng.comm_connected = False
        # every rank must take the same route: one that failed sends all of them to the host-side sum over the store
        all_ok = all(grp.allgather(bool(ok)))
        if not all_ok:
            eng.comm_connected = False
        allreduce_via = ("rccl (amc_allreduce_sum: one ncclAllReduce on the engine's communication stream)" if all_ok else
                         "launcher's TCP store (host-side sum: RCCL communicator unavailable)")
        grp.barrier()
    # what this process really runs on, for a line that verifies itself: the communicator's own rank count (ncclCommCount),
    # the librccl and HIP runtime files bound (under the launcher torch is imported first, so libamc.so binds torch's
    # bundled ROCm; a bare single-process run binds /opt/rocm)
    rt = A.runtime_info()
    ci = eng.comm_info() if getattr(eng, "comm_connected", False) else None
    stack = {"torch_imported": "torch" in sys.modules, "rccl_library_forced": A.comm_library_forced(),
             "hip_runtime_version": rt["hip_runtime_version"], "hip_runtime": rt["hip_runtime"],
             "rccl_ranks": None if ci is None else ci["n_ranks"], "rccl_rank": None if ci is None else ci["rank"],
             "rccl_version": None if ci is None else ci["rccl_version"], "librccl": None if ci is None else ci["librccl"]}
    if grp is not None:
        stacks = grp.allgather(stack)
        counts = [st["rccl_ranks"] for st in stacks]
        # every rank's communicator must report the whole world; anything else is shown, not hidden
        stack["rccl_ranks"] = None if any(c is None for c in counts) else min(counts)
        stack["rccl_ranks_by_rank"] = counts
        stack["hip_runtime_versions_by_rank"] = [st["hip_runtime_version"] for st in stacks]
        stack["hip_runtimes_by_rank"] = sorted({st["hip_runtime"] for st in stacks})
        stack["torch_imported"] = any(st["torch_imported"] for st in stacks)
    cb_every = CALLBACK_EVERY_MULTI if grp is not None else 0
    if os.environ.get("AMC_BENCH_CB_EVERY"):                 # developer knob: separate the cost of the callbacks from the process group's
        cb_every = int(os.environ["AMC_BENCH_CB_EVERY"])

    pending = [False]

    def finish_callback():
        """merge the callback sums enqueued one period ago over the shards (the host never drains the sweep queue): the shards'
        exact integer records through ONE ncclAllReduce used as a gather (amc_allreduce_xsum) -- callback_energy and
        callback_acceptance then have the same bits whatever N is."""
        if pending[0]:
            pending[0] = False
            rec, steps = eng.reduce_end_exact()
            return eng.reduce_records_value(sharding.allreduce_xsum(rec, eng), steps)
        return None

    def run_steps(n):
        """n steps of the loop: one launch per sweep, queued in stretches by ONE engine call each (amc_sweep_launches: the same n
        launches, minus the per-launch crossing of the language boundary -- a 20-step timed region is 0.6 ms, and a host hiccup
        of a few tens of microseconds between two launches is a few per cent of it); with callbacks, every cb_every-th step is
        the sweep whose state the callbacks observe (sums formed in-kernel), the previous callback's sums read just before it."""
        if not cb_every:
            eng.sweep_launches(n)
            return
        i = 0
        while i < n:
            plain = min(n - i, cb_every - 1 - (i % cb_every))
            if plain > 0:
                eng.sweep_launches(plain)
                i += plain
            if i < n:
                finish_callback()
                eng.sweep_reduce_begin(1)
                pending[0] = True
                i += 1

    def local_sync():
        eng.sync()                           # everything this rank has queued is done (sweeps: engine's stream; the callback sums' all-reduce is host-synchronous)

    def barrier():
        """torch.cuda.synchronize() + barrier of the contract: this rank's queue drained, then all ranks meet -- in ONE
        tiny all-reduce of the engines' communicator where there is one (tens of microseconds), over the launcher's TCP
        store otherwise (milliseconds: a 20-step timed region is 0.6 ms)."""
        local_sync()
        if grp is not None:
            if getattr(eng, "comm_connected", False):
                eng.allreduce_sum([0.0])
            else:
                grp.barrier()

    rccl_failed = False
    t_spin = time.perf_counter()                 # clock ramp (untimed), then the W warm-up steps
    spin_blocks = []                             # ms per 200-launch block of the ramp
    while True:
        b0 = time.perf_counter()
        # with callbacks: the very steps of the timed loop, callbacks included -- whatever W is (the driver's W = 5 never
        # reaches a callback step), the first launch of the sum-forming kernel form and the first real all-reduces happen here;
        # so does a one-off of the stack: some 0.1-0.2 s into a process's first callback-bearing steps the queue stands still
        # once for 30-50 ms (one gap in the kernel trace, with or without a communicator; profiles/NOTES_r03.md)
        run_steps(200)
        finish_callback()
        eng.sync()
        spin_blocks.append((time.perf_counter() - b0) * 1e3)
        # steady state, not a fixed time: the clock keeps coming up for as long as the host leg before it kept the GPU idle
        # (BENCH_r04: 20 timed steps at 32.7 us behind a fixed 0.6 s, the same run's later blocks at 29.5), so the ramp ends
        # when the last three blocks agree within 1 % -- not before --spinup-s, not after --spinup-cap-s
        spun = time.perf_counter() - t_spin
        last = spin_blocks[-3:]
        steady = len(last) == 3 and max(last) <= 1.01 * min(last)
        time_up = (spun >= args.spinup_s and steady) or spun >= args.spinup_cap_s
        if grp is None:
            if time_up:
                break
        # every rank must leave after the same number of collectives: the ranks agree through one more of them
        elif sharding.all_ranks(time_up, eng):
            break
    # ... and then in the timed region's own rhythm: untimed blocks of the very bracket that follows (barrier, K steps, drain), so that
    # the timed block is one more block of a steady sequence and not the first of its kind after the ramp's 200-launch stretches
    # (the two HIP-event records that bracket the timed launches cost it ~1.5 % at K = 20 against the blocks of `repeat`, which carry
    # none: measured by leaving them out, three runs each way on one box; they stay -- roofline.frac_events is taken over the timed region)
    for _ in range(max(0, args.preblocks)):
        barrier()
        run_steps(args.steps)
        finish_callback()
        local_sync()
    run_steps(args.warmup)
    finish_callback()
    barrier()
    eng.timing_begin()
    t0 = time.perf_counter()
    run_steps(args.steps)
    finish_callback()                      # the last callback's all-reduce belongs to the timed region
    eng.timing_mark()                      # end event behind the K-th launch (asynchronous) ...
    local_sync()
    elapsed = time.perf_counter() - t0     # this rank's K steps, done; the MAX over ranks below is the job's time
    barrier()                              # the closing barrier of the bracket (its own latency is no part of any rank's steps)
    event_ms = eng.timing_end()            # ... HIP events on the engine's stream, bracketing exactly the K launches
    if grp is not None:
        both = grp.allgather((elapsed, event_ms))              # max over ranks
        elapsed, event_ms = max(b[0] for b in both), max(b[1] for b in both)

    # the same K-step block, repeated (not part of `value`): a 20-step driver run is 0.7 ms of timed region, and one slow
    # launch moves it by 5 %
    # ... and, with --min-gpu-seconds, for at least that long: the device stays busy long enough for a sampler outside this
    # process to see it at work (every block is the timed region's own: same steps, same barriers)
    rep_ms = []
    t_rep = time.perf_counter()
    while True:
        n_done = len(rep_ms)
        more = n_done < max(0, args.repeats) or (args.repeats > 0 and time.perf_counter() - t_rep < args.min_gpu_seconds)
        if grp is not None:
            more = sharding.all_ranks(more, eng) if args.min_gpu_seconds > 0 else n_done < max(0, args.repeats)
        if not more:
            break
        barrier()
        r0 = time.perf_counter()
        run_steps(args.steps)
        finish_callback()
        local_sync()
        rep_ms.append((time.perf_counter() - r0) * 1e3 / args.steps)
    if grp is not None and rep_ms:
        rep_ms = [max(col) for col in zip(*grp.allgather(rep_ms))]

    rec, steps_counted = eng.reduce_exact()
    red = eng.reduce_records_value(sharding.allreduce_xsum(rec, eng), steps_counted)
    n = red[3]
    energy, acceptance = red[0] / n, red[4] / n

    # the sweepstep = 16 form, reported separately -- the state stays in registers between the 16 MH steps of a
    # launch, so bytes per update fall to 1 and a fraction of the HBM roofline means nothing there
    fused = None
    if world == 1:
        for _ in range(5):
            eng.sweep(16)
        eng.sync()
        fts = []
        for _ in range(3):
            eng.timing_begin()
            for _ in range(20):
                eng.sweep(16)
            fts.append(eng.timing_end() * 1e3 / (20 * 16))
        fused = {"mh_steps_per_launch": 16, "us_per_sweep_min": min(fts), "us_per_sweep_median": sorted(fts)[1],
                 "chain_updates_per_s": (stop - start) / (min(fts) * 1e-6),
                 "note": "one launch = 16 mc_step! per chain with x in registers (amc_sweep(h, 16)); not the headline: no HBM "
                         "round trip per sweep, roofline fraction not meaningful"}

    ladder_rows = None
    if rank == 0 and world == 1 and not args.no_ladder and args.chains_per_gpu == M_PER_GPU:
        ladder_rows = ladder(A, (4 * M_PER_GPU, 16 * M_PER_GPU), local_rank)
    others = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        others = other_configs(A, stop - start, local_rank)
    if rank == 0:
        prof = pmc_profile()
        updates = m_global * args.steps
        launch_s = event_ms * 1e-3 / args.steps
        # ONE clock per line: `achieved` / `frac` come from the interval `value` and `ms_per_step` come from (this rank's K
        # steps by the host clock, max over ranks), so frac x peak x ms_per_step IS the algorithmic bytes of a launch; the
        # HIP-event figure of the same K launches (the kernel's own duration, what rocprofv3's kernel trace averages) rides
        # beside it as achieved_events / frac_events / avg_launch_us
        achieved = BYTES_PER_UPDATE * (stop - start) * args.steps / elapsed / 1e9
        achieved_events = BYTES_PER_UPDATE * (stop - start) / launch_s / 1e9
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
                "callbacks_allreduce_every": cb_every, "callbacks_allreduce_via": allreduce_via,
                **stack,
                "sharding": "contiguous global chain ids per rank; no data-path collective",
                "multi_gpu_note": "N > 1: callback sums merged over the shards as exact integer records by the engines' own RCCL "
                                  "communicator (amc_allreduce_xsum: one ncclAllReduce used as a gather), unique id and barriers over a "
                                  "plain-socket store (no torch in the worker); rccl_ranks is ncclCommCount of that communicator "
                                  "(null: no communicator); this repository's own GPU runs have one device, so N > 1 has never "
                                  "been run by the builder",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "clock": "host perf_counter around the K timed steps, max over ranks: the interval of `value` and `ms_per_step`",
                "achieved_events": achieved_events, "frac_events": achieved_events / HBM_PEAK_GBS,
                "events_note": "HIP events on the engine's stream around the same K launches (avg_launch_us = their mean): the kernel's own "
                               "duration, comparable with the rocprofv3 kernel-trace average under profiles/",
                # the committed PMC figures were taken at 1e7 chains per launch: quoted only for that size
                "traffic": (prof or {}).get("traffic") if stop - start == M_PER_GPU else None,
                "valu_busy": (prof or {}).get("valu_busy") if stop - start == M_PER_GPU else None,
                "traffic_provenance": None if prof is None else {
                    "measured_in_this_run": False, "source": prof["source"], "commit": prof["commit"],
                    "kernel_sources_unchanged_since": prof["kernel_sources_unchanged_since"],
                    "note": "rocprofv3 PMC passes need the profiler around the process: static figures from the committed profile"},
                "kernel": "amc::sweep_kernel<harmonic, K=1, pool-wide counter>",
                "algorithmic_bytes_per_launch": BYTES_PER_UPDATE * (stop - start),
                "avg_launch_us": launch_s * 1e6,
                "regime": regime_of(stop - start),
                "ladder": ladder_rows,
                "note": "f64 VALU-bound in practice (Philox + Box-Muller + exp per update), see DESIGN.md §6",
            },
            "repeat": None if not rep_ms else {
                "blocks": len(rep_ms), "steps_per_block": args.steps, "ms_per_step_min": min(rep_ms),
                "ms_per_step_median": sorted(rep_ms)[len(rep_ms) // 2],
                "ms_per_step_p10": sorted(rep_ms)[len(rep_ms) // 10], "ms_per_step_p90": sorted(rep_ms)[(9 * len(rep_ms)) // 10],
                "ms_per_step_all": rep_ms[:5] if len(rep_ms) > 16 else rep_ms,        # the first five of a long series
                "seconds": sum(rep_ms) * args.steps * 1e-3},
            "check": {"mean_energy": energy, "acceptance": acceptance},
            "fused_sweepstep16": fused,
            "other_configs": others,
        }
        if cpu_line is not None:
            result["cpu_baseline"] = cpu_line
        # a stand-in or a site build in place of librccl: the line says so, and carries no `value` unless a test asked for it
        if stack.get("rccl_library_forced") and os.environ.get("AMC_BENCH_ALLOW_FORCED_RCCL") != "1":
            result["value"] = None
            result["value_withheld"] = "AMC_RCCL_LIBRARY replaced librccl in this run: not a measurement over RCCL"
        # N > 1: the line is a measurement over RCCL or it is no measurement -- a run in which some rank fell back to the
        # launcher's store for the sums, or whose communicator does not span all N ranks, keeps its diagnostics and loses `value`
        # (a fresh process is the retry, never a re-exec)
        if grp is not None and world > 1 and not (getattr(eng, "comm_connected", False) and stack.get("rccl_ranks") == world):
            result["value"] = None
            result["value_withheld"] = (f"RCCL did not carry this run (communicator over {stack.get('rccl_ranks')} of {world} ranks; "
                                        f"callback sums via: {allreduce_via}): not a measurement of the N-GPU path")
            rccl_failed = True
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    eng.close()
    if grp is not None:
        grp.barrier()
        # every rank leaves with the same code: rank 0 decided, the others learn it over the store
        if bool(grp.broadcast(bool(rccl_failed) if rank == 0 else None)):
            raise SystemExit(3)


if __name__ == "__main__":
    main()
