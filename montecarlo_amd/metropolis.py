"""``Metropolis``: the HIP-backed algorithm object for the reference's hot path.

Mirror of ``Metropolis`` (src/metropolis.jl:232-291) and its ``make_step!`` (:302-309) behind the
plugin protocol, plus the two callbacks that read its state (``callback_acceptance`` :319-321,
``callback_energy`` example/particle_1d/particle_1d.jl:68-70).  All arithmetic runs in
libamc.so's kernels; this file only owns the shard bookkeeping and the cross-shard sum.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import numpy as np

from . import sharding
from ._capi import AMC_RED_HEADER, HipEngine, SplitEngine
from .simulation import AriannaAlgorithm, Simulation, _calls, julia_repr
from .system import Move, ParticleChains


GAUSS_SAMPLE = "sigma*z"
GAUSS_LOGQ = "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0"      # particle_1d.jl:52-54
GAUSS_DLOGQ = "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma"


def _move_signature(m):
    """What the kernels must be compiled for to serve this move: its (policy, action) type with their expressions."""
    act = (m.action.perform, m.action.invert) if hasattr(m.action, "perform") else (None, None)
    if hasattr(m.policy, "logq"):
        return ("script", m.policy.sample, m.policy.logq, m.policy.dlogq) + act
    scale = getattr(m.policy, "scale", None)
    if scale is not None:
        return ("scaled", scale) + act
    return ("gauss",) + act


def _class_expressions(sig):
    """(sample, logq, dlogq, perform, invert) of one class of a mixed pool; the Gaussian policies written out as expressions."""
    if sig[0] == "script":
        return tuple(sig[1:6])
    if sig[0] == "scaled":
        w = f"(sigma*({sig[1]}))"
        return (f"{w}*z", f"-(delta*delta)/(2.0*({w}*{w})) - amc_log(6.283185307179586*({w}*{w}))/2.0",
                f"((delta*delta)/({w}*{w}*{w}))*({sig[1]}) - ({sig[1]})/{w}", sig[2], sig[3])
    return (GAUSS_SAMPLE, GAUSS_LOGQ, GAUSS_DLOGQ, sig[1], sig[2])


class Metropolis(AriannaAlgorithm):
    """Metropolis(chains; pool, sweepstep=1, seed=1, ...) -- src/metropolis.jl:288-291.

    ``R`` / ``parallel`` are accepted for signature compatibility and ignored: the generator is
    the counter-based Philox of DESIGN.md §3 and the chains run on the GPU.
    ``per_chain_counters``: keep ``pools[c][k].accepted_calls / total_calls`` PER CHAIN on the device (``download_counters``).
    Default: only when the pool has more than one move (``callback_acceptance`` is then a mean of per-chain ratios); for a
    single move every chain has the same ``total_calls``, the pool-wide accepted total gives the same callback and
    ``Move`` totals, and the sweep is ~20 % faster without the per-chain step log.
    ``streams``: > 1 splits this rank's shard into that many sub-shards on separate HIP streams of the same GPU, so that
    the launch boundary of one overlaps the body of another (~8 % faster back-to-back single-sweep launches with 2 at
    1e7 chains -- visible only where the step loop is not host-bound, which a Python loop with a callback at every
    step is); per-chain results are unchanged (global chain ids), the policy-gradient path then sums on the host.
    ``engine_factory`` is a TEST SEAM (default and only shipped engine: ``HipEngine``); tests on
    CPU-only boxes pass a double built on the oracle to exercise this host logic.
    """

    fusable = True

    def __init__(self, chains: ParticleChains, pool: Optional[Sequence[Move]] = None, sweepstep: int = 1,
                 seed: int = 1, R=None, parallel: bool = False, device: Optional[int] = None,
                 per_chain_counters: Optional[bool] = None, download_on_finalise: bool = True,
                 engine_factory: Optional[Callable[..., object]] = None, streams: int = 1, **extras):
        if pool is None or len(pool) == 0:
            raise ValueError("Metropolis: pool is missing")
        if not all(isinstance(m, Move) for m in pool):
            raise TypeError("Metropolis: pool must hold Move objects")
        self.pool = tuple(pool)
        self.sweepstep = int(sweepstep)
        self.seed = int(seed)
        self.parallel = bool(parallel)
        self.chains = chains
        self.rank, self.world_size = sharding.world()
        start, stop = sharding.shard_range(len(chains), self.rank, self.world_size)
        if stop <= start:
            raise ValueError(f"rank {self.rank} owns no chains: use at most {(len(chains) + 1) // 2} ranks")
        self.shard = (start, stop)
        chains.shard = (start, stop)
        self.download_on_finalise = download_on_finalise
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        factory = engine_factory or HipEngine
        if engine_factory is None and int(streams) > 1:
            # sub-shards of this rank's shard on separate streams (SplitEngine): their launches overlap
            factory = lambda **kw: SplitEngine(n_parts=int(streams), **kw)
        extra = {} if getattr(chains, "reward", None) is None else {"reward_expr": chains.reward}
        if getattr(chains, "dtype", "f64") != "f64":
            extra["dtype"] = chains.dtype
        n_params = 1
        kinds = [_move_signature(m) for m in self.pool]
        if len(set(kinds)) > 1:
            # A pool that MIXES policy / action types -- every Move carries its own (src/metropolis.jl:140-162): one expression set
            # ("class") per distinct (policy, action) pair, the built-in Gaussian displacement written out as expressions too
            # (amc_create_mixed_model).  One parameter per move.
            if any(int(getattr(m.policy, "n_params", 1)) != 1 for m in self.pool):
                raise ValueError("a pool that mixes policy types takes policies of one parameter")
            order = list(dict.fromkeys(kinds))
            if len(order) > 4:
                raise ValueError("a pool may mix at most 4 different (policy, action) types")
            classes = [_class_expressions(k) for k in order]
            if any(c[2] is None for c in classes):            # d logq / d sigma for every class or for none (no estimator then)
                classes = [(c[0], c[1], None, c[3], c[4]) for c in classes]
            extra["classes"] = classes
            extra["class_of_move"] = [order.index(k) for k in kinds]
        elif kinds[0][0] == "scaled":
            extra["scale_expr"] = kinds[0][1]                  # one policy expression per handle: the kernels are compiled for it
        elif kinds[0][0] == "script":
            extra["proposal"] = kinds[0][1:]
            # a policy with several parameters (ScriptPolicy(n_params=P)): the engine takes one parameter vector per move
            n_params = int(getattr(self.pool[0].policy, "n_params", 1))
            if n_params > 1:
                pr = extra["proposal"]
                extra["n_params"] = n_params
                extra["proposal"] = pr[:2] + (None if pr[2] is None else list(pr[2]),) + pr[3:]
        self.n_params = n_params
        self.engine = factory(n_chains=stop - start, chain_offset=start, n_chains_global=len(chains), **extra,
                              potential=chains.potential, beta=chains.beta,
                              sigma=[(m.parameters.copy() if n_params > 1 else m.sigma) for m in self.pool], weight=[m.weight for m in self.pool],
                              seed=self.seed, sweepstep=self.sweepstep,
                              per_chain_counters=bool(per_chain_counters) or len(self.pool) > 1, device=device)
        # sharded on GPUs: the engine gets an RCCL communicator of its own (callback sums and the estimator's fold are then
        # ONE ncclAllReduce on its stream); otherwise (one rank, or a CPU test double) sums go through sharding.allreduce_sum
        self._comm_connected = self.world_size > 1 and sharding.connect_engine(self.engine)
        self._epoch = 0          # bumped whenever the device state changes
        self._red_key = None
        self._red_val = None
        # Reductions callbacks hold whose sums still sit in the engine, oldest first.  The engine keeps up to two reductions in
        # flight and hands them back oldest first (amc_reduce_begin / amc_reduce_end): while the device forms the sums of one
        # observation point the host reads those of the one before.
        self._inflight = []

    # ---- plugin protocol ---------------------------------------------------------------
    def initialise(self, simulation: Simulation) -> None:
        if getattr(self, "restored", False):        # storage.restore() already put the state on the device
            self._epoch += 1
            return
        start, stop = self.shard
        ch = self.chains
        beta = None if ch.beta_array is None else ch.beta_array[start:stop]
        if ch.x is not None:
            x = ch.x if ch.x.shape[0] == stop - start else ch.x[start:stop]
            self.engine.upload_state(x, beta)
        else:
            lo, hi = ch.init_uniform if ch.init_uniform is not None else (0.0, 0.0)
            if beta is not None:
                self.engine.upload_state(np.zeros(stop - start), beta)
            self.engine.init_uniform(lo, hi)
        self._epoch += 1

    def make_step(self, simulation: Simulation, with_reductions: bool = False) -> None:
        """make_step!(simulation, ::Metropolis), src/metropolis.jl:302-309: one sweep of every chain.

        ``with_reductions``: a callback observes the state right after this sweep (run() looks ahead in the
        schedule), so the sums are formed inside the sweep launch instead of by a second pass over the chains."""
        self._drop_pending_reduction()
        if with_reductions and hasattr(self.engine, "sweep_reduce_begin"):
            self._settle_claimed(keep=1)        # two reductions in flight per engine: the last callback's may still be on its
            self.engine.sweep_reduce_begin(1)   # way, the one before it is fetched now -- two callback periods after it was queued
            self._pending_red_epoch = self._epoch + 1
        else:
            self.engine.sweep(1)
        self._epoch += 1

    def _drop_pending_reduction(self) -> None:
        if getattr(self, "_pending_red_epoch", None) is not None:
            self._settle_claimed()              # the engine hands reductions back oldest first: fetch what callbacks hold,
            self.engine.reduce_end_exact()      # then discard the one nobody asked for
            self._pending_red_epoch = None

    def _settle_claimed(self, keep: int = 0) -> None:
        """Fetch the oldest reductions callbacks hold until at most `keep` of them are left in the engine."""
        while len(self._inflight) > keep:
            self._inflight[0].result()

    def make_steps(self, simulation: Simulation, n: int) -> None:
        """n consecutive make_step!s fused in one launch (state stays in registers)."""
        self._drop_pending_reduction()
        self.engine.sweep(n)
        self._epoch += 1

    def finalise(self, simulation: Simulation) -> None:
        self._drop_pending_reduction()
        self._settle_claimed()
        self.set_reduction_needs(None)          # the run's narrowing ends with the run: a caller of reductions() gets every entry
        if getattr(self, "device_params_dirty", False):
            self.pull_parameters()
        if self.download_on_finalise:
            x, e = self.engine.download_state(want_e=True)
            self.chains.x, self.chains.e = x, e        # this rank's shard, like chains[c].x / .e
        acc, tot = self.engine.counter_totals()
        tot_all = sharding.allreduce_sum(np.concatenate([acc, tot]).astype(np.float64), self.engine)
        K = len(self.pool)
        for k, move in enumerate(self.pool):
            move.accepted_calls = int(tot_all[k])
            move.total_calls = int(tot_all[K + k])

    def sync(self) -> None:
        self.engine.sync()

    def write_algorithm(self, io, scheduler) -> None:
        """src/metropolis.jl:346-363."""
        io.write("\tMetropolis\n")
        io.write(f"\t\tCalls: {_calls(scheduler)}\n")
        io.write(f"\t\tMC steps per simulation step: {self.sweepstep}\n")
        io.write(f"\t\tSeed: {self.seed}\n")
        io.write(f"\t\tParallel: gfx950 HIP kernels, {self.world_size} shard(s)\n")
        io.write("\t\tMoves:\n")
        for k, move in enumerate(self.pool, start=1):
            io.write(f"\t\t\tMove {k}:\n\t\t\t\tAction: Displacement\n\t\t\t\tPolicy: StandardGaussian\n")
            io.write(f"\t\t\t\tParameters: {julia_repr(move.parameters)}\n\t\t\t\tWeight: {julia_repr(move.weight)}\n")

    # ---- state the dependants read (pools, parameters) ------------------------------------
    def set_parameters(self, k: int, parameters) -> None:
        """Push Move.parameters of move k to the device copy (after learning_step!, update.jl:53)."""
        p = np.atleast_1d(np.asarray(parameters, dtype=np.float64))
        self.engine.set_parameters(k, p)
        if self.pool[k].parameters is not parameters:
            self.pool[k].parameters[...] = p

    def pull_parameters(self) -> None:
        """Refresh the host copies of Move.parameters from the device (after device-resident learning steps)."""
        for k, move in enumerate(self.pool):
            move.parameters[...] = self.engine.get_parameters(k)
        self.device_params_dirty = False

    def parameters_async(self, t: int) -> "ParameterRead":
        """The parameters of every move as of the steps queued so far, as a ticket (engine.parameters_begin: a copy in stream
        order; ``result()`` fetches it).  The engine keeps ONE such read in flight: algorithms that ask at the same time step
        share the ticket, and a new time step first fetches the older read -- whoever holds it gets the values all the same."""
        tk = getattr(self, "_param_ticket", None)
        if tk is not None and tk.t == t:
            return tk
        if tk is not None:
            tk.result()
        tk = ParameterRead(self.engine, t)
        self._param_ticket = tk
        return tk

    def download_counters(self):
        """pools[c][k].accepted_calls / total_calls of this rank's shard, shape (K, M_local)."""
        return self.engine.download_counters()

    # ---- reductions behind the callbacks ---------------------------------------------------
    def reductions(self) -> dict:
        """One device reduction + ONE all-reduce per observation point, shared by all callbacks."""
        return self.reductions_async().result()

    def reductions_async(self) -> "Reduction":
        """The reduction of the CURRENT state as a ticket: ``result()`` fetches the sums (and all-reduces them over the
        shards) when somebody needs the numbers.  The engine forms them in stream order -- inside the sweep launch that
        produced the state when run() saw the callback coming -- so a caller that asks late (StoreCallbacks writes a
        callback's row when the next one is due) never makes the host wait for the device, and the sweeps queued in
        between are not held back by it.  Values are those of the state at the time of this call either way."""
        key = self._epoch
        cols = getattr(self, "_red_cols", 7)
        if self._red_key == key and self._red_val is not None and (self._red_val.cols & cols) == cols:
            return self._red_val        # (a cached reduction formed with fewer columns than are asked for now is not reused)
        ticket = Reduction(self, cols)
        if getattr(self, "_pending_red_epoch", None) == self._epoch:
            self._pending_red_epoch = None      # formed inside the launch (make_step(with_reductions=True)): claim it
            self._inflight.append(ticket)
        else:
            self._drop_pending_reduction()
            self._settle_claimed(keep=1)
            self.engine.reduce_begin()
            self._inflight.append(ticket)
        self._red_key, self._red_val = key, ticket
        return ticket

    def _fetch(self, ticket: "Reduction") -> None:
        while self._inflight and self._inflight[0] is not ticket:       # oldest first
            self._inflight[0].result()
        assert self._inflight and self._inflight[0] is ticket
        self._inflight.pop(0)
        ticket._finish(*self.engine.reduce_end_exact())

    def set_reduction_needs(self, needs) -> None:
        """What the callbacks of this run read of a reduction (run() collects it from the StoreCallbacks' callbacks: names among
        "energy", "mean_x", "mean_x2", "acceptance"; None: everything).  The engine then forms only those sums over x --
        callback_energy (particle_1d.jl:68-70) and callback_acceptance (metropolis.jl:319-321), the reference's own two,
        need sum e alone; what nobody asked for reads NaN."""
        if not hasattr(self.engine, "set_reduce_columns"):
            return
        cols = 7 if needs is None else ((1 if "energy" in needs else 0) | (2 if "mean_x" in needs else 0) | (4 if "mean_x2" in needs else 0))
        self._drop_pending_reduction()
        self.engine.set_reduce_columns(cols)
        if cols != getattr(self, "_red_cols", 7):
            # a reduction cached for the current state was formed with the old columns: a caller who asks after the change
            # (callback_moments(simulation) after a run narrowed to energy) gets a new one, not NaN entries
            self._red_key = self._red_val = None
        self._red_cols = cols

    def invalidate_reductions(self) -> None:
        """Called by algorithms that move the chains behind Metropolis' back (the estimator)."""
        self._epoch += 1


class ParameterRead:
    """A queued read of the moves' parameters (see Metropolis.parameters_async)."""

    def __init__(self, engine, t: int):
        engine.parameters_begin()
        self._engine, self.t, self._val = engine, t, None

    def result(self) -> np.ndarray:
        if self._val is None:
            self._val = self._engine.parameters_end()
        return self._val


class Reduction:
    """The callback sums of one observation point (see Metropolis.reductions_async)."""

    def __init__(self, metropolis: Metropolis, cols: int = 7):
        self._met = metropolis
        self._val = None
        self.cols = cols                # the sums over x it was formed with (bit 0 energy, 1 mean_x, 2 mean_x2)

    def _finish(self, records: np.ndarray, steps_counted: int) -> None:
        # the shards' partial sums are merged as exact integer records (reproducible sums, include/amc.h) and rounded ONCE:
        # the rows a callback writes do not depend on how many GPUs the chains are spread over
        merged = sharding.allreduce_xsum(records, self._met.engine)
        red = self._met.engine.reduce_records_value(merged, steps_counted)
        n = red[3]
        self._val = {
            "energy": red[0] / n,                      # mean(system.e for system in chains)
            "mean_x": red[1] / n,
            "mean_x2": red[2] / n,
            "n_chains": int(round(n)),
            "acceptance": red[AMC_RED_HEADER:] / n,    # mean over chains of accepted/total per move
        }

    def result(self) -> dict:
        if self._val is None:
            self._met._fetch(self)
        return self._val


def _find_metropolis(simulation: Simulation):
    cached = simulation.__dict__.get("_the_metropolis")
    if cached is not None:
        return cached
    found = [a for a in simulation.algorithms if isinstance(a, Metropolis)]
    if len(found) != 1:
        # the reference's generator splat only works with exactly one Metropolis (metropolis.jl:320)
        raise ValueError(f"callbacks need exactly one Metropolis in the algorithm list, found {len(found)}")
    simulation.__dict__["_the_metropolis"] = found[0]
    return found[0]


def _deferrable(pick, needs):
    """A callback f(simulation) over the engine's reduction, plus ``f.deferred(simulation)``: the same value as a thunk over
    the reduction ticket of the current state, for StoreCallbacks to evaluate when it writes the row; ``f.needs``: the entries
    of the reduction it reads (Metropolis.set_reduction_needs)."""
    def wrap(fn):
        def deferred(simulation: Simulation):
            ticket = _find_metropolis(simulation).reductions_async()
            return lambda: pick(ticket.result())
        fn.deferred = deferred
        fn.needs = tuple(needs)
        return fn
    return wrap


@_deferrable(lambda r: float(r["energy"]), needs=("energy",))
def callback_energy(simulation: Simulation) -> float:
    """callback_energy, example/particle_1d/particle_1d.jl:68-70: mean energy over all chains."""
    return float(_find_metropolis(simulation).reductions()["energy"])


@_deferrable(lambda r: np.array(r["acceptance"], dtype=np.float64), needs=("acceptance",))
def callback_acceptance(simulation: Simulation) -> np.ndarray:
    """callback_acceptance, src/metropolis.jl:319-321: per move, mean over chains of
    accepted_calls/total_calls (NaN before the first step, like the reference's 0/0)."""
    return np.array(_find_metropolis(simulation).reductions()["acceptance"], dtype=np.float64)


@_deferrable(lambda r: np.array([r["mean_x"], r["mean_x2"]]), needs=("mean_x", "mean_x2"))
def callback_moments(simulation: Simulation) -> np.ndarray:
    """[mean(x), mean(x^2)] over all chains: the statistic test/distribution_test.jl:36-37 checks."""
    r = _find_metropolis(simulation).reductions()
    return np.array([r["mean_x"], r["mean_x2"]])
