"""Caller side of the hot path: the algorithm plugin protocol, ``Simulation`` and ``run``.

Python mirror of the reference's driver so the HIP ``Metropolis`` slots in the way an
``AriannaAlgorithm`` does in Julia:
  protocol      src/algorithms.jl:6-37   initialise / make_step! / finalise / write_algorithm
  Simulation    src/simulation.jl:16-88  (NamedTuple list -> instances, dependencies by type)
  schedules     src/simulation.jl:95-117 build_schedule (three methods)
  run!          src/simulation.jl:175-204
  StoreCallbacks src/algorithms.jl:62-109, StoreParameters src/metropolis.jl:380-450,
  PrintTimeSteps src/algorithms.jl:310-323
Per-chain text I/O (StoreTrajectories / StoreBackups / StoreLastFrames, src/algorithms.jl:154-303) lives in
trajectories.py (the reference's file layout, for ensembles small enough to have one file per chain); storage.py holds
what replaces it at M = 1e7.  One addition over the reference: ``run(fuse=True)`` looks ahead in
the schedules and issues one fused launch for every stretch of sweeps no other algorithm
observes (results identical to stepping one by one).
"""
from __future__ import annotations

import math
import os
import sys
import time
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import sharding


class AriannaAlgorithm:
    """src/algorithms.jl:6-37: the four generic functions every algorithm implements."""

    def initialise(self, simulation: "Simulation") -> None:
        return None

    def make_step(self, simulation: "Simulation") -> None:
        return None

    def finalise(self, simulation: "Simulation") -> None:
        return None

    def write_algorithm(self, io, scheduler) -> None:
        io.write(f"\t{type(self).__name__}\n")


# ---------------------------------------------------------------------------------------
# build_schedule, src/simulation.jl:95-117
# ---------------------------------------------------------------------------------------
def build_schedule(steps: int, burn: int, spec) -> List[int]:
    """Three methods, dispatched on the type of ``spec`` like the reference:
    int dt -> burn:dt:steps U [steps]; float base -> log-spaced; list block -> repeated blocks."""
    steps, burn = int(steps), int(burn)
    if isinstance(spec, bool):
        raise TypeError("build_schedule: spec must be int, float or list")
    if isinstance(spec, (int, np.integer)):
        out = list(range(burn, steps + 1, int(spec)))
        if not out or out[-1] != steps:
            out.append(steps)
        return out
    if isinstance(spec, float):
        nmax = math.floor(math.log(steps - burn) / math.log(spec))
        vals = [burn]
        for n in range(nmax + 1):
            p = spec ** n
            if p != math.floor(p):
                raise ValueError(f"InexactError: Int({p})")   # Int(base^n) throws in the reference
            vals.append(burn + int(p))
        vals.append(steps)
        return _unique(vals)
    block = [int(b) for b in spec]
    nblock = (steps - burn) // block[-1]
    vals: List[int] = []
    for m in range(1, nblock + 1):
        vals.extend(b + burn + (m - 1) * block[-1] for b in block)
    vals.append(steps)
    return [v for v in _unique(vals) if v <= steps]


def _unique(vals: Iterable[int]) -> List[int]:
    seen, out = set(), []
    for v in vals:
        if v not in seen:
            seen.add(v)
            out.append(v)
    return out


# ---------------------------------------------------------------------------------------
# Julia-style printing of callback values ("$t $(callback(simulation))", algorithms.jl:99)
# ---------------------------------------------------------------------------------------
def julia_repr(v) -> str:
    """string(v) as Julia prints Float64 / Vector{Float64}: shortest round-trip digits, fixed
    notation for 1e-4 <= |x| < 1e6, otherwise d.ddde±n; NaN, Inf; vectors as [a, b]."""
    if isinstance(v, (list, tuple, np.ndarray)):
        return "[" + ", ".join(julia_repr(x) for x in np.asarray(v).tolist()) + "]"
    if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
        return str(int(v))
    x = float(v)
    if 1e-4 <= abs(x) < 1e6:
        return repr(x)              # same shortest round-trip digits, same fixed notation (the common case, fast)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    sign = "-" if math.copysign(1.0, x) < 0 else ""
    if x == 0.0:
        return sign + "0.0"
    m, _, e = repr(abs(x)).partition("e")
    ip, _, fp = m.partition(".")
    digits = ip + fp
    point = len(ip) + (int(e) if e else 0)          # value = 0.digits * 10^point
    stripped = digits.lstrip("0")
    point -= len(digits) - len(stripped)
    digits = stripped.rstrip("0") or "0"
    e10 = point - 1
    if -5 < e10 < 6:
        if point <= 0:
            body = "0." + "0" * (-point) + digits
        elif point >= len(digits):
            body = digits + "0" * (point - len(digits)) + ".0"
        else:
            body = digits[:point] + "." + digits[point:]
    else:
        body = digits[0] + "." + (digits[1:] or "0") + "e" + str(e10)
    return sign + body


# ---------------------------------------------------------------------------------------
# Simulation, src/simulation.jl:16-88
# ---------------------------------------------------------------------------------------
class Simulation:
    def __init__(self, chains, algorithm_list: Sequence[Dict[str, Any]], steps: int, path: str = "data",
                 verbose: bool = False):
        self.chains = chains
        self.steps = int(steps)
        self.t = 0
        self.path = path
        self.verbose = verbose
        self.rank, self.world_size = sharding.world()
        schedulers, algorithms, names = [], [], []
        for constructor in algorithm_list:
            constructor = dict(constructor)
            cls = constructor.pop("algorithm")
            names.append(cls)
            scheduler = constructor.pop("scheduler", range(1, self.steps + 1))      # :74
            dependencies = constructor.pop("dependencies", None)
            kwargs = dict(constructor)
            if dependencies is not None:                                             # :77-81
                parents = [a for a, n in zip(algorithms, names[:-1]) if n in tuple(dependencies)]
                kwargs["dependencies"] = tuple(parents)
            kwargs.update(path=path, steps=self.steps, verbose=verbose)              # :82
            algorithms.append(cls(chains, **kwargs))                                 # :83
            schedulers.append(scheduler)
        self.algorithms = tuple(algorithms)
        self.schedulers = tuple(schedulers)
        assert len(self.schedulers) == len(self.algorithms)                         # :45
        for s in self.schedulers:
            if not isinstance(s, range):
                assert all(0 <= x <= self.steps for x in s), "scheduler entries must lie in [0, steps]"   # :46
                assert all(a <= b for a, b in zip(s, s[1:])), "scheduler must be sorted"              # :47
        # counters = findfirst(x -> x > 0, scheduler)  (:49), 0-based here; None if nothing > 0
        self.counters: List[Optional[int]] = [next((i for i, x in enumerate(s) if x > 0), None)
                                              for s in self.schedulers]
        if self.rank == 0:
            os.makedirs(path, exist_ok=True)                                         # :50

    # summary.log, src/simulation.jl:124-165 (rank 0 only)
    def _write_summary(self) -> None:
        if self.rank != 0:
            return
        with open(os.path.join(self.path, "summary.log"), "w") as f:
            f.write("SIMULATION SUMMARY\n\nSimulation:\n")
            f.write(f"\tSteps: {self.steps}\n\tNumber of chains: {len(self.chains)}\n")
            f.write(f"\tNumber of algorithms: {len(self.algorithms)}\n\tVerbose: {str(self.verbose).lower()}\n")
            f.write(f"\tStarted on {time.strftime('%Y-%m-%dT%H:%M:%S')}\n\nSystem:\n")
            tname = "Float32" if getattr(self.chains, "dtype", "f64") == "f32" else "Float64"
            f.write(f"\tParticle{{{tname}}} x {len(self.chains)} ({self.chains.potential}, SoA on MI355X)\n\n")
            f.write("Algorithms:\n")
            for alg, sched in zip(self.algorithms, self.schedulers):
                alg.write_algorithm(f, sched)
            f.write("\n")

    def _update_summary(self, sim_time: float) -> None:
        if self.rank != 0:
            return
        with open(os.path.join(self.path, "summary.log"), "a") as f:
            f.write(f"Report:\n\tSimulation time: {sim_time} s\n")

    def _finalise_summary(self) -> None:
        if self.rank != 0 or not os.path.exists(os.path.join(self.path, "summary.log")):
            return
        total = 0
        for root, _dirs, files in os.walk(self.path):
            total += sum(os.path.getsize(os.path.join(root, fn)) for fn in files)
        with open(os.path.join(self.path, "summary.log"), "a") as f:
            f.write(f"\tSimulation size: {total / 1024 ** 2} MB\n")
            f.write(f"\tStatus: Completed on {time.strftime('%Y-%m-%dT%H:%M:%S')}\n")


def _calls(scheduler) -> int:
    """write_algorithm's "Calls:" line: length(filter(x -> 0 < x <= scheduler[end], scheduler))."""
    if len(scheduler) == 0:
        return 0
    last = scheduler[-1]
    return sum(1 for x in scheduler if 0 < x <= last)


def _due(simulation: Simulation, k: int) -> Optional[int]:
    c = simulation.counters[k]
    s = simulation.schedulers[k]
    if c is None or c >= len(s):
        return None          # the reference would index out of bounds here (simulation.jl:186)
    return s[c]


def _declare_reduction_needs(sim: Simulation) -> None:
    """Tell the sampler which entries of a reduction this run's consumers read, so that the launches which form the sums form
    those and no others.  Consumers are the algorithms marked ``wants_reductions`` (StoreCallbacks -- callbacks are plain
    functions f(simulation) in the reference, src/algorithms.jl:97-102; the engine-backed ones carry ``needs`` --,
    StoreHistogram); one that does not say what it reads (``reduction_needs()`` missing or None: any user function among the
    callbacks) keeps everything, and so does a run without consumers (a caller of Metropolis.reductions()) or with an
    algorithm of a type this package does not define: a user's algorithm may call Metropolis.reductions() or a callback
    inside its make_step, and must not read NaN for a sum nobody declared."""
    needs, known, any_consumer = set(), True, False
    for alg in sim.algorithms:
        if not type(alg).__module__.startswith(__name__.rsplit(".", 1)[0] + "."):
            known = False
        if not getattr(alg, "wants_reductions", False):
            continue
        any_consumer = True
        n = alg.reduction_needs() if hasattr(alg, "reduction_needs") else None
        if n is None:
            known = False
        else:
            needs.update(n)
    for alg in sim.algorithms:
        if hasattr(alg, "set_reduction_needs"):
            alg.set_reduction_needs(needs if (any_consumer and known) else None)


def run(simulation: Simulation, fuse: bool = True) -> None:
    """run!(simulation), src/simulation.jl:175-204.

    ``fuse``: when the next n time steps schedule nothing but one fusable algorithm (the HIP
    Metropolis), issue them as ONE launch.  Algorithm order inside a time
    step and every observable state are the same as without fusion.
    """
    sim = simulation
    try:
        _declare_reduction_needs(sim)
        for alg in sim.algorithms:                                                  # :179-181
            alg.initialise(sim)
        sim._write_summary()                                                        # :182
        t_start = time.perf_counter()
        t = 1
        n_alg = len(sim.algorithms)
        while t <= sim.steps:                                                       # :184
            sim.t = t
            advance = 1
            due = [k for k in range(n_alg) if _due(sim, k) == t]
            if fuse and len(due) == 1 and getattr(sim.algorithms[due[0]], "fusable", False):
                k = due[0]
                others = [d for j in range(n_alg) if j != k for d in [_due(sim, j)] if d is not None]
                horizon = min(others) if others else sim.steps + 1
                n = _consecutive(sim.schedulers[k], sim.counters[k], t, min(horizon - 1, sim.steps))
                if n > 1:
                    sim.algorithms[k].make_steps(sim, n)
                    sim.counters[k] += n
                    advance = n
                    sim.t = t + n - 1
                    t += advance
                    continue
            if fuse and len(due) >= 2:
                # [Metropolis, estimator(, update)] as ONE engine call for as many time steps as schedule nothing else --
                # and through the step at which later algorithms of the list (callbacks, StoreParameters) are due as well:
                # they run after the three at that t (list order, src/simulation.jl:185-190) and see the same state
                m, n = _pgmc_group(sim, due, t)
                if n:
                    for k in due[:m]:
                        sim.counters[k] += n
                    sim.t = t + n - 1
                    for k in range(n_alg):                                          # the others due at the group's last step
                        if k not in due[:m] and _due(sim, k) == sim.t:
                            sim.algorithms[k].make_step(sim)
                            sim.counters[k] += 1
                    t += n
                    continue
            for i, k in enumerate(due):                                             # :185-190
                alg = sim.algorithms[k]
                if getattr(alg, "fusable", False) and _observed_next(sim, due[i + 1:]):
                    alg.make_step(sim, with_reductions=True)    # callbacks follow at this t: sums formed in-kernel
                else:
                    alg.make_step(sim)
                sim.counters[k] += 1
            t += advance
        for alg in sim.algorithms:
            sync = getattr(alg, "sync", None)
            if sync:
                sync()
        sim_time = time.perf_counter() - t_start
        if sim.verbose and sim.rank == 0:
            print(f"\nSimulation completed in {sim_time} s")
        sim._update_summary(sim_time)                                               # :193
    finally:
        for alg in sim.algorithms:                                                  # :196-198
            alg.finalise(sim)
        sim._finalise_summary()


def _pgmc_group(simulation: Simulation, due: Sequence[int], t: int):
    """If the algorithms due at ``t`` START with [fusable Metropolis, its device-resident estimator(, that estimator's
    update)] in this order, issue the next n time steps with that pattern as ONE engine call and return (m, n): m = how
    many leading entries of ``due`` the pattern covers.  The group runs up to and including the first step at which
    any other algorithm is due, provided every algorithm due there comes AFTER the pattern in the list (it then runs
    after the group and observes the state it leaves, exactly as when stepping one by one); otherwise it stops the step
    before.  (0, 0) when the pattern does not apply."""
    sim = simulation
    algs = [sim.algorithms[k] for k in due]
    if not getattr(algs[0], "fusable", False) or getattr(algs[1], "pgmc_role", None) != "estimator":
        return 0, 0
    est = algs[1]
    if getattr(est, "metropolis", None) is not algs[0] or not hasattr(est, "make_steps_grouped"):
        return 0, 0
    m, upd = 2, None
    if len(due) >= 3 and getattr(algs[2], "pgmc_role", None) == "update" and getattr(algs[2], "estimator", None) is est:
        m, upd = 3, algs[2]
    head = list(due[:m])
    others = [(j, d) for j in range(len(sim.algorithms)) if j not in head for d in [_due(sim, j)] if d is not None]
    s_other = min((d for _, d in others), default=sim.steps + 1)
    after = all(j > head[-1] for j, d in others if d == s_other)
    t_last = min(s_other if after else s_other - 1, sim.steps)
    n = min(_consecutive(sim.schedulers[k], sim.counters[k], t, t_last) for k in head)
    if n < 1:
        return 0, 0
    # a callback scheduled at the group's last step observes the state the group leaves: its sums ride in the last launch
    observed = after and t + n - 1 == s_other and _observed_next(sim, sorted(j for j, d in others if d == s_other))
    return (m, n) if est.make_steps_grouped(sim, n, upd, with_reductions=observed) else (0, 0)


def _observed_next(simulation: Simulation, later: Sequence[int]) -> bool:
    """True if, among the algorithms still due at this time step, one that reads the reductions comes before
    any that moves the chains (the estimator perturbs x: src/PolicyGuided/gradients.jl:98,103)."""
    for k in later:
        alg = simulation.algorithms[k]
        if getattr(alg, "mutates_chains", False):
            return False
        if getattr(alg, "wants_reductions", False):
            return True
    return False


def _consecutive(scheduler, counter: int, t: int, t_last: int) -> int:
    """How many of t, t+1, ... (<= t_last) are consecutive entries of ``scheduler`` from ``counter``."""
    if t_last < t:
        return 0
    if isinstance(scheduler, range) and scheduler.step == 1:
        return max(0, min(t_last, scheduler[-1]) - t + 1)
    n = 0
    while counter + n < len(scheduler) and scheduler[counter + n] == t + n and t + n <= t_last:
        n += 1
    return n


# ---------------------------------------------------------------------------------------
# StoreCallbacks, src/algorithms.jl:62-109
# ---------------------------------------------------------------------------------------
class StoreCallbacks(AriannaAlgorithm):
    """StoreCallbacks (src/algorithms.jl:62-109): one file per callback, one row "$t $(callback(simulation))" per scheduled t.

    ``defer`` (default on): when every callback of the list knows how to hand out its value as a thunk (``f.deferred``: the
    engine-backed callback_energy / callback_acceptance / callback_moments), the row of time t is WRITTEN when the next
    scheduled time comes (or at finalise): the sums are formed on the device in stream order at t, and the host reads
    them a callback period later instead of stalling the sweep loop for them.  Rows, values and their order in the files
    are the same; only ``rows`` / the files lag one entry behind during the run."""

    wants_reductions = True

    def __init__(self, chains, path=None, callbacks=None, store_first: bool = True, store_last: bool = False,
                 defer: bool = True, **extras):
        self.callbacks = tuple(callbacks or ())
        self.store_first, self.store_last = store_first, store_last
        self.defer = bool(defer) and len(self.callbacks) > 0 and all(hasattr(cb, "deferred") for cb in self.callbacks)
        self._pending = None                # (t, [thunk per callback]) of the row not written yet
        self.rank, _ = sharding.world()
        self.paths = [os.path.join(path, cb.__name__.replace("callback_", "") + ".dat") for cb in self.callbacks]
        self.files: List[Any] = []
        self.rows: List[List[tuple]] = [[] for _ in self.callbacks]   # in-memory copy (t, value)

    def initialise(self, simulation: Simulation) -> None:
        if self.rank == 0:
            for p in self.paths:
                os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
            self.files = [open(p, "w") for p in self.paths]
        if self.store_first:                                                        # :93
            self.make_step(simulation)

    def reduction_needs(self):
        """The entries of a reduction the callbacks read (their ``needs``), or None when one of them does not say."""
        out = set()
        for cb in self.callbacks:
            n = getattr(cb, "needs", None)
            if n is None:
                return None
            out.update(n)
        return out

    def _write(self, t: int, values) -> None:
        for i, value in enumerate(values):
            self.rows[i].append((t, value))
            if self.rank == 0:
                self.files[i].write(f"{t} {julia_repr(value)}\n")
                self.files[i].flush()

    def flush(self) -> None:
        """Write the row still held back (deferred mode)."""
        if self._pending is not None:
            t, thunks = self._pending
            self._pending = None
            self._write(t, [thunk() for thunk in thunks])     # every rank evaluates: the reduction all-reduces

    def make_step(self, simulation: Simulation) -> None:                            # :97-102
        if self.defer:
            thunks = [cb.deferred(simulation) for cb in self.callbacks]    # claims the reduction of the state at this t
            self.flush()                                                     # the previous row: its sums are long there
            self._pending = (simulation.t, thunks)
            return
        self._write(simulation.t, [cb(simulation) for cb in self.callbacks])   # every rank calls: the callback all-reduces

    def finalise(self, simulation: Simulation) -> None:
        if self.store_last:
            self.make_step(simulation)
        self.flush()
        for f in self.files:
            f.close()
        self.files = []

    def write_algorithm(self, io, scheduler) -> None:
        io.write("\tStoreCallbacks\n")
        io.write(f"\t\tCalls: {_calls(scheduler)}\n")
        io.write(f"\t\tCallbacks: {[cb.__name__ for cb in self.callbacks]}\n")
        io.write(f"\t\tStore first: {str(self.store_first).lower()}\n\t\tStore last: {str(self.store_last).lower()}\n")


# ---------------------------------------------------------------------------------------
# StoreParameters, src/metropolis.jl:380-450
# ---------------------------------------------------------------------------------------
class StoreParameters(AriannaAlgorithm):
    """StoreParameters(chains; dependencies=(Metropolis,), ids, store_first, store_last) (src/metropolis.jl:380-450): one row
    ``t parameters`` per scheduled time and stored move.

    ``defer`` (default): when the parameters live on the device (learning steps taken there, PolicyGradientUpdate), the read of
    time t is QUEUED at t (engine.parameters_begin: a copy in stream order) and its row is written when the next scheduled
    time comes, or at finalise -- the same rows in the same files, one entry late during the run.  Reading at once
    (defer=False) makes the host wait for every queued step and leaves the queue empty behind it: 6.5 us per time step of the
    reference's PGMC script at 1e7 chains (parameters stored on the callbacks' schedule)."""

    def __init__(self, chains, dependencies=None, path=None, ids=None, store_first: bool = True,
                 store_last: bool = False, defer: bool = True, **extras):
        assert dependencies is not None and len(dependencies) == 1
        metropolis = dependencies[0]
        pool = metropolis.pool
        self.metropolis = metropolis
        self.ids = list(range(len(pool))) if ids is None else list(ids)
        self.parameters_list = [pool[k].parameters for k in self.ids]
        self.store_first, self.store_last = store_first, store_last
        self.rank, _ = sharding.world()
        self.paths = [os.path.join(path, "parameters", str(k + 1), "parameters.dat") for k in self.ids]  # 1-based dirs
        self.files: List[Any] = []
        self.rows: List[List[tuple]] = [[] for _ in self.ids]
        self.defer = bool(defer)
        self._pending = None                     # (t, ticket) of the read queued on the engine whose row is not written yet

    def initialise(self, simulation: Simulation) -> None:
        if self.rank == 0:
            for p in self.paths:
                os.makedirs(os.path.dirname(p), exist_ok=True)
            self.files = [open(p, "w") for p in self.paths]
        if self.store_first:
            self.make_step(simulation)

    def _write(self, t: int, values) -> None:
        for i, prm in enumerate(values):
            self.rows[i].append((t, prm))
            if self.rank == 0:
                self.files[i].write(f"{t} {julia_repr(prm)}\n")
                self.files[i].flush()

    def flush(self) -> None:
        """Fetch the queued read, if any, and write its row."""
        if self._pending is None:
            return
        (t, ticket), self._pending = self._pending, None
        sigma = ticket.result()
        values = []
        for k, prm in zip(self.ids, self.parameters_list):
            v = prm.copy()
            v[...] = sigma[k]
            values.append(v)
        self._write(t, values)

    def make_step(self, simulation: Simulation) -> None:
        met = self.metropolis
        if getattr(met, "device_params_dirty", False):
            if self.defer and hasattr(met.engine, "parameters_begin"):
                self.flush()                        # the previous row first (its read was queued a period ago)
                # sigma as of the steps queued so far; one read in flight per engine, shared by every StoreParameters of this
                # Metropolis that is due at the same t (Metropolis.parameters_async)
                self._pending = (simulation.t, met.parameters_async(simulation.t))
                return
            met.pull_parameters()                   # sigma was updated by a device-resident learning step
        self.flush()
        self._write(simulation.t, [prm.copy() for prm in self.parameters_list])

    def finalise(self, simulation: Simulation) -> None:
        self.flush()
        if self.store_last:
            if getattr(self.metropolis, "device_params_dirty", False):
                self.metropolis.pull_parameters()
            self._write(simulation.t, [prm.copy() for prm in self.parameters_list])
        for f in self.files:
            f.close()
        self.files = []

    def write_algorithm(self, io, scheduler) -> None:
        io.write("\tStoreParameters\n")
        io.write(f"\t\tCalls: {_calls(scheduler)}\n")


class PrintTimeSteps(AriannaAlgorithm):
    """src/algorithms.jl:310-323: progress bar."""

    def __init__(self, chains, **extras):
        self.rank, _ = sharding.world()

    def make_step(self, simulation: Simulation) -> None:
        if self.rank != 0:
            return
        t = simulation.t
        percent = t / simulation.steps
        filled = int(round(percent * 50))
        bar = "\033[1;34m" + "■" * filled + "\033[0m" + "□" * (50 - filled)
        sys.stdout.write(f"\rProgress: [{bar}] {percent * 100:.0f}% t = {t}")
        sys.stdout.flush()
