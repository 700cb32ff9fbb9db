"""Device-side stand-ins for the reference's per-chain text I/O, usable at 10^7 chains.

The reference writes one text file per chain (StoreTrajectories src/algorithms.jl:154-210, StoreBackups
:264-303, StoreLastFrames :221-251; row format "$t $(x)", example/particle_1d/particle_1d.jl:63-66) and can
not resume (no loader; RNG state and move counters are not saved).  What those files are consumed for is
served here without 10^7 open files:

  StoreHistogram   pooled-position histogram over the schedule (the density plot of
                   MC_harmonic_oscillator.jl:40-51 and the mean/std check of test/distribution_test.jl:33-37)
  StoreSnapshots   strided binary snapshots x[first::stride] at the scheduled times (a trajectory subset)
  checkpoint / restore   complete, exact resume: positions, per-chain counters, sigma and the two Philox
                   step indices -- the generator is counter-based, so (seed, step) IS the RNG state.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from . import sharding
from .metropolis import Metropolis
from .simulation import AriannaAlgorithm, Simulation, _calls


class StoreHistogram(AriannaAlgorithm):
    """StoreHistogram(chains; dependencies=(Metropolis,), lo, hi, bins, scheduler): accumulates the histogram
    of all chain positions at every scheduled time; `finalise` all-reduces it and rank 0 writes
    `histogram.dat` (bin_lo bin_hi count) plus the pooled mean / std estimated from the bin-free moments."""

    wants_reductions = True

    def __init__(self, chains, dependencies=None, path=None, lo: float = -2.0, hi: float = 2.0, bins: int = 200,
                 **extras):
        assert dependencies is not None and len(dependencies) == 1 and isinstance(dependencies[0], Metropolis)
        self.metropolis: Metropolis = dependencies[0]
        self.lo, self.hi, self.bins = float(lo), float(hi), int(bins)
        self.counts = np.zeros(self.bins + 3, dtype=np.uint64)
        self.moments = np.zeros(3)                 # n, sum x, sum x^2 over all samples (global)
        self.path = os.path.join(path, "histogram.dat")
        self.rank, _ = sharding.world()

    def reduction_needs(self):
        return ("mean_x", "mean_x2")          # the bin-free moments (run() tells the sampler: simulation._declare_reduction_needs)

    def _settle(self) -> None:
        """The moments of the previous sample (a reduction ticket claimed then) are fetched when the next one is due."""
        ticket, self._ticket = getattr(self, "_ticket", None), None
        if ticket is not None:
            r = ticket.result()
            n = r["n_chains"]
            self.moments += np.array([n, r["mean_x"] * n, r["mean_x2"] * n])

    def make_step(self, simulation: Simulation) -> None:
        eng = self.metropolis.engine
        self._settle()
        # The engine keeps ONE running histogram.  The first StoreHistogram of a Metropolis to sample owns it (its counts stay
        # on the device until finalise and its moments are read one sample late: nothing makes the host wait for the queued
        # sweeps); any other instance -- same bins or not -- fetches its counts at each sample time.
        owner = getattr(self.metropolis, "_histogram_owner", None)
        if owner is None and hasattr(eng, "histogram_accumulate"):
            owner = self.metropolis._histogram_owner = self
        on_device = owner is self
        if on_device:
            eng.histogram_accumulate(self.lo, self.hi, self.bins)
            self._on_device = True
        if on_device:
            self._ticket = self.metropolis.reductions_async()
        else:
            self.counts += eng.histogram(self.lo, self.hi, self.bins)
            self._ticket = self.metropolis.reductions_async()
            self._settle()

    def finalise(self, simulation: Simulation) -> None:
        self._settle()
        if getattr(self, "_on_device", False):
            self.counts += self.metropolis.engine.histogram_fetch(self.bins, reset=True)
            self._on_device = False
            self.metropolis._histogram_owner = None
        total = sharding.allreduce_sum(self.counts.astype(np.float64))
        self.global_counts = np.rint(total).astype(np.uint64)
        n, sx, sxx = self.moments
        self.mean = sx / n if n else float("nan")
        self.std = float(np.sqrt(max(sxx / n - self.mean ** 2, 0.0))) if n else float("nan")
        if self.rank == 0:
            os.makedirs(os.path.dirname(self.path), exist_ok=True)
            edges = self.lo + (self.hi - self.lo) * np.arange(self.bins + 1) / self.bins
            with open(self.path, "w") as f:
                f.write(f"# samples {int(n)} mean {self.mean!r} std {self.std!r} below {int(self.global_counts[self.bins])} "
                        f"above {int(self.global_counts[self.bins + 1])} nan {int(self.global_counts[self.bins + 2])}\n")
                for i in range(self.bins):
                    f.write(f"{edges[i]!r} {edges[i + 1]!r} {int(self.global_counts[i])}\n")

    def write_algorithm(self, io, scheduler) -> None:
        io.write(f"\tStoreHistogram\n\t\tCalls: {_calls(scheduler)}\n\t\tRange: [{self.lo}, {self.hi}) in {self.bins} bins\n")


class StoreSnapshots(AriannaAlgorithm):
    """StoreSnapshots(chains; dependencies=(Metropolis,), stride, scheduler): rows (t, x[first::stride]) of this
    rank's shard, kept in memory and written as `snapshots_rank<r>.npy` at finalise (binary trajectory subset)."""

    def __init__(self, chains, dependencies=None, path=None, stride: int = 1000, max_chains: int = 4096,
                 store_first: bool = True, **extras):
        assert dependencies is not None and len(dependencies) == 1 and isinstance(dependencies[0], Metropolis)
        self.metropolis: Metropolis = dependencies[0]
        start, stop = self.metropolis.shard
        self.stride = int(stride)
        first_global = -(-start // self.stride) * self.stride          # first multiple of stride inside the shard
        self.first = first_global - start
        self.count = 0 if self.first >= stop - start else min(int(max_chains), (stop - start - 1 - self.first) // self.stride + 1)
        self.chain_ids = start + self.first + self.stride * np.arange(self.count)
        self.times: List[int] = []
        self.rows: List[np.ndarray] = []
        self.store_first = store_first
        self.rank, _ = sharding.world()
        self.path = os.path.join(path, f"snapshots_rank{self.rank}.npy")

    def initialise(self, simulation: Simulation) -> None:
        if self.store_first:
            self.make_step(simulation)

    def make_step(self, simulation: Simulation) -> None:
        self.times.append(simulation.t)
        self.rows.append(self.metropolis.engine.download_strided(self.first, self.stride, self.count))

    def finalise(self, simulation: Simulation) -> None:
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        data = np.column_stack([np.array(self.times, dtype=np.float64), np.array(self.rows).reshape(len(self.times), self.count)])
        np.save(self.path, data)
        np.save(self.path.replace(".npy", "_chain_ids.npy"), self.chain_ids)

    def write_algorithm(self, io, scheduler) -> None:
        io.write(f"\tStoreSnapshots\n\t\tCalls: {_calls(scheduler)}\n\t\tChains: every {self.stride}th ({self.count} on rank {self.rank})\n")


def checkpoint(metropolis: Metropolis, path: str, estimator=None) -> str:
    """Write this rank's shard so that `restore` continues the run bit for bit."""
    eng = metropolis.engine
    if getattr(metropolis, "device_params_dirty", False):
        metropolis.pull_parameters()
    if estimator is not None:
        estimator.refresh()
    x, _ = eng.download_state(want_e=False)
    start, stop = metropolis.shard
    acc_tot, tot_tot = eng.counter_totals()
    data = dict(x=x, shard=np.array([start, stop]), n_chains_global=len(metropolis.chains), seed=metropolis.seed,
                sweepstep=metropolis.sweepstep, step=eng.step, estimator_step=eng.estimator_step,
                sigma=np.array([m.sigma for m in metropolis.pool]), weight=np.array([m.weight for m in metropolis.pool]),
                accepted_total=acc_tot, total_total=tot_tot)
    try:
        acc, tot = eng.download_counters()
        data.update(accepted=acc, total=tot)
    except Exception:
        pass                                       # pool-wide counter mode: totals above are the state
    if metropolis.chains.beta_array is not None:
        data["beta"] = metropolis.chains.beta_array[start:stop]
    if estimator is not None:
        data["gd"] = np.array([[g.j, g.grad_j[0], g.grad_logq_forward[0], g.g[0, 0], g.n] for g in estimator.gradients_data])
    os.makedirs(path, exist_ok=True)
    fn = os.path.join(path, f"checkpoint_rank{metropolis.rank}.npz")
    np.savez(fn, **data)
    return fn


def restore(metropolis: Metropolis, path: str, estimator=None) -> None:
    """Load `checkpoint`'s file into a freshly constructed Metropolis with the same pool/seed/sharding."""
    d = np.load(os.path.join(path, f"checkpoint_rank{metropolis.rank}.npz"))
    start, stop = metropolis.shard
    if list(d["shard"]) != [start, stop] or int(d["seed"]) != metropolis.seed or int(d["n_chains_global"]) != len(metropolis.chains):
        raise ValueError("checkpoint does not match this Metropolis (shard / seed / ensemble size)")
    eng = metropolis.engine
    eng.upload_state(d["x"], d["beta"] if "beta" in d else None)
    for k, s in enumerate(d["sigma"]):
        metropolis.set_parameters(k, [float(s)])
    eng.step = int(d["step"])
    eng.estimator_step = int(d["estimator_step"])
    if "accepted" in d:
        eng.upload_counters(d["accepted"], d["total"])
    else:
        eng.set_counter_totals(int(d["accepted_total"][0]), int(d["total_total"][0]) // (stop - start))
    if estimator is not None and "gd" in d:
        from .policy_guided import GradientData
        estimator.gradients_data = [GradientData(float(r[0]), np.array([r[1]]), np.array([r[2]]), np.array([[r[3]]]), int(r[4]))
                                    for r in d["gd"]]
        if estimator.device_resident and estimator.learn_ids:
            # the running sums live in the engine (amc_pg_accumulate adds to them there): without this the first update
            # after the resume would average the post-resume samples only
            if not hasattr(eng, "pg_set_accumulated"):
                if any(int(r[4]) != 0 for r in d["gd"]):
                    raise ValueError("restore: this engine cannot take over non-empty device-resident gradients_data")
            else:
                eng.pg_set_accumulated(estimator.learn_ids, np.asarray(d["gd"], dtype=np.float64))
    metropolis.chains.x = None                      # initialise() must not overwrite the restored state
    metropolis.restored = True
    metropolis.invalidate_reductions()
