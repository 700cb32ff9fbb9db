"""The reference's per-chain text I/O, in its own file layout, for ensembles small enough to have one file per chain.

    StoreTrajectories   src/algorithms.jl:154-210   path/trajectories/<c>/trajectory.dat, one row per scheduled t
    StoreLastFrames     src/algorithms.jl:221-251   path/trajectories/<c>/lastframe.dat, written at finalise
    StoreBackups        src/algorithms.jl:264-303   path/trajectories/<c>/restart_t<t>.dat, one file per scheduled t

Row formats: DAT is particle_1d's ``store_trajectory`` (example/particle_1d/particle_1d.jl:63-66), ``"$t $(system.x)"``;
TXT falls through to the generic method (src/algorithms.jl:186-189), ``"$t, $system"``, i.e. Julia's ``show`` of the
``Particle`` struct.  Chain directories are 1-based global chain ids, as ``eachindex(chains)`` numbers them.

The chains live on the device, so a scheduled call is one download of the selected positions (and energies for TXT).
The reference opens M files; that is its design for M ~ 10, not for 10^7: above ``max_chains`` (default 4096) the
constructor refuses unless a strided selection ``select=(first, stride, count)`` is given -- `storage.StoreSnapshots`
and `storage.StoreHistogram` are the tools for large ensembles.  Like the reference these algorithms need no
``dependencies``: they find the one Metropolis of the simulation (the reference reads ``simulation.chains`` directly).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np

from .metropolis import _find_metropolis
from .simulation import AriannaAlgorithm, Simulation, _calls, julia_repr


@dataclass(frozen=True)
class DAT:
    """DAT <: Format (src/algorithms.jl:135-140)."""
    extension: str = ".dat"


@dataclass(frozen=True)
class TXT:
    """TXT <: Format (src/algorithms.jl:123-128)."""
    extension: str = ".txt"


def _repr_state(v: float, dtype: str) -> str:
    """string(x) for x::Float64 or x::Float32 (shortest round-trip digits of the type; Float32 exponents print as f)."""
    if dtype != "f32":
        return julia_repr(v)
    f = np.float32(v)
    if np.isnan(f):
        return "NaN32"
    if np.isinf(f):
        return "Inf32" if f > 0 else "-Inf32"
    a = abs(float(f))
    if f == 0 or 1e-4 <= a < 1e6:
        return np.format_float_positional(f, unique=True, trim="0")
    m, e = np.format_float_scientific(f, unique=True, trim="0").split("e")
    return f"{m}f{int(e)}"


class _PerChainFiles(AriannaAlgorithm):
    def __init__(self, chains, path=None, fmt=None, max_chains: int = 4096,
                 select: Optional[Tuple[int, int, int]] = None,
                 row: Optional[Callable[[int, float, float, float], str]] = None, **extras):
        if path is None:
            raise ValueError(f"{type(self).__name__} needs path=")
        self.fmt = fmt if fmt is not None else DAT()
        # store_trajectory(io, system, t, fmt) is the method a model overrides (src/algorithms.jl:182-189;
        # particle_1d.jl:63-66 does): `row(t, x, beta, e)` returns the line (without the newline) for one chain
        self.row = row
        self.chains = chains
        self.root = os.path.join(path, "trajectories")
        self.select = select
        self.max_chains = int(max_chains)
        self.metropolis = None
        self.local_first = self.local_stride = self.local_count = 0
        self.ids = np.zeros(0, dtype=np.int64)

    # the selection in this rank's shard; resolved at first use because Metropolis owns the sharding
    def _bind(self, simulation: Simulation) -> None:
        if self.metropolis is not None:
            return
        self.metropolis = _find_metropolis(simulation)
        start, stop = self.metropolis.shard
        if self.select is None:
            if len(self.chains) > self.max_chains:
                raise ValueError(f"{type(self).__name__}: {len(self.chains)} chains would open one file each; pass "
                                 "select=(first, stride, count), raise max_chains, or use StoreSnapshots / StoreHistogram")
            first, stride, count = 0, 1, len(self.chains)
        else:
            first, stride, count = (int(v) for v in self.select)
            if first < 0 or stride < 1 or count < 0 or (count and first + (count - 1) * stride >= len(self.chains)):
                raise ValueError("select=(first, stride, count) leaves the ensemble")
        ids = first + stride * np.arange(count, dtype=np.int64)
        ids = ids[(ids >= start) & (ids < stop)]
        self.ids = ids
        self.local_count = len(ids)
        self.local_first = int(ids[0] - start) if len(ids) else 0
        self.local_stride = stride
        self.dirs = [os.path.join(self.root, str(int(c) + 1)) for c in ids]         # "$c", 1-based
        for d in self.dirs:
            os.makedirs(d, exist_ok=True)

    def _rows(self, simulation: Simulation):
        """The text of one row per selected chain at the current t."""
        if self.local_count == 0:
            return []
        eng = self.metropolis.engine
        dtype = getattr(self.chains, "dtype", "f64")
        t = simulation.t
        if self.row is not None:
            xs, es = eng.download_state(want_e=True)
            start, _ = self.metropolis.shard
            beta = self.chains.beta_array
            sel = self.local_first + self.local_stride * np.arange(self.local_count)
            return [self.row(t, float(xs[i]), float(self.chains.beta if beta is None else beta[start + i]), float(es[i])) + "\n"
                    for i in sel]
        if isinstance(self.fmt, DAT):
            x = eng.download_strided(self.local_first, self.local_stride, self.local_count)
            return [f"{t} {_repr_state(v, dtype)}\n" for v in x]
        # generic store_trajectory: "$t, $system" -> Particle{Float64}(x, beta, e)
        xs, es = eng.download_state(want_e=True)
        sel = self.local_first + self.local_stride * np.arange(self.local_count)
        start, _ = self.metropolis.shard
        beta = self.chains.beta_array
        tname = "Float32" if dtype == "f32" else "Float64"

        def show(v):          # `show` of a field inside the struct: Float32 literals carry their f0 / f-5 suffix
            r = _repr_state(v, dtype)
            return r if dtype != "f32" or "f" in r or r.endswith("32") else r + "f0"

        rows = []
        for i in sel:
            b = self.chains.beta if beta is None else beta[start + i]
            if dtype == "f32":
                b = float(np.float32(b))
            rows.append(f"{t}, Particle{{{tname}}}({show(xs[i])}, {show(b)}, {show(es[i])})\n")
        return rows


class StoreTrajectories(_PerChainFiles):
    """StoreTrajectories(chains; path, fmt=DAT(), store_first=true, store_last=false), src/algorithms.jl:154-210."""

    def __init__(self, chains, path=None, fmt=None, store_first: bool = True, store_last: bool = False,
                 flush: bool = False, **extras):
        super().__init__(chains, path=path, fmt=fmt, **extras)
        self.store_first, self.store_last, self.flush = bool(store_first), bool(store_last), bool(flush)
        self.files = []

    def initialise(self, simulation: Simulation) -> None:
        self._bind(simulation)
        self.files = [open(os.path.join(d, "trajectory" + self.fmt.extension), "w") for d in self.dirs]
        if self.store_first:
            self.make_step(simulation)

    def make_step(self, simulation: Simulation) -> None:
        for f, row in zip(self.files, self._rows(simulation)):
            f.write(row)
            if self.flush:                 # the reference flushes every row (:201); off by default: one syscall per chain and t
                f.flush()

    def finalise(self, simulation: Simulation) -> None:
        if self.store_last:
            self.make_step(simulation)
        for f in self.files:
            f.close()
        self.files = []

    def write_algorithm(self, io, scheduler) -> None:
        io.write(f"\tStoreTrajectories\n\t\tCalls: {_calls(scheduler)}\n")


class StoreLastFrames(_PerChainFiles):
    """StoreLastFrames(chains; path, fmt=DAT()), src/algorithms.jl:221-251: the state at the end of the run."""

    def initialise(self, simulation: Simulation) -> None:
        self._bind(simulation)

    def make_step(self, simulation: Simulation) -> None:
        return None

    def finalise(self, simulation: Simulation) -> None:
        self._bind(simulation)
        for d, row in zip(self.dirs, self._rows(simulation)):
            with open(os.path.join(d, "lastframe" + self.fmt.extension), "w") as f:
                f.write(row)

    def write_algorithm(self, io, scheduler) -> None:
        io.write("\tStoreLastFrames\n")


class StoreBackups(_PerChainFiles):
    """StoreBackups(chains; path, fmt=DAT(), store_first=false, store_last=false), src/algorithms.jl:264-303:
    restart_t<t> files in the reference's format.  (For a resume that continues bit for bit use storage.checkpoint.)"""

    def __init__(self, chains, path=None, fmt=None, store_first: bool = False, store_last: bool = False, **extras):
        super().__init__(chains, path=path, fmt=fmt, **extras)
        self.store_first, self.store_last = bool(store_first), bool(store_last)

    def initialise(self, simulation: Simulation) -> None:
        self._bind(simulation)
        if self.store_first:
            self.make_step(simulation)

    def make_step(self, simulation: Simulation) -> None:
        self._bind(simulation)
        name = f"restart_t{simulation.t}{self.fmt.extension}"
        for d, row in zip(self.dirs, self._rows(simulation)):
            with open(os.path.join(d, name), "w") as f:
                f.write(row)

    def finalise(self, simulation: Simulation) -> None:
        if self.store_last:
            self.make_step(simulation)

    def write_algorithm(self, io, scheduler) -> None:
        io.write(f"\tStoreBackups\n\t\tCalls: {_calls(scheduler)}\n")
