"""Policy-guided Monte Carlo on the HIP path: src/PolicyGuided/ of the reference.

  GradientData, +, average      gradients.jl:41-85
  PolicyGradientEstimator       estimator.jl:38-134   (per-sample arithmetic: HIP kernel K3)
  PolicyGradientUpdate          update.jl:14-57
  Static VPG BLPG BLAPG NPG ANPG BLANPG + learning_step!   learning.jl:16-164
The per-chain x per-sample estimate (gradients.jl:93-121) runs on the GPU; the fold over chains
is the kernel's block reduction + ONE all-reduce; the O(P^2) optimiser algebra stays on the host
(as in the reference) and the new sigma is pushed to the device copy.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import sharding
from .metropolis import Metropolis
from .simulation import AriannaAlgorithm, Simulation, _calls


# ---- optimisers, learning.jl ---------------------------------------------------------------
class PolicyGradient:
    """abstract type PolicyGradient (learning.jl:9)."""


@dataclass(frozen=True)
class Static(PolicyGradient):
    pass


@dataclass(frozen=True)
class VPG(PolicyGradient):
    eta: float


@dataclass(frozen=True)
class BLPG(PolicyGradient):
    eta: float


@dataclass(frozen=True)
class BLAPG(PolicyGradient):
    delta: float
    eps_id: float = 0.0


@dataclass(frozen=True)
class NPG(PolicyGradient):
    eta: float
    eps_id: float = 0.0


@dataclass(frozen=True)
class ANPG(PolicyGradient):
    delta: float
    eps_id: float = 0.0


@dataclass(frozen=True)
class BLANPG(PolicyGradient):
    delta: float
    eps_id: float = 0.0


@dataclass
class GradientData:
    """gradients.jl:41-47: objective j, its gradient, grad logq of the forward action, metric g, count n."""
    j: float
    grad_j: np.ndarray
    grad_logq_forward: np.ndarray
    g: np.ndarray
    n: int

    def __add__(self, o: "GradientData") -> "GradientData":                       # gradients.jl:68-76
        return GradientData(self.j + o.j, self.grad_j + o.grad_j, self.grad_logq_forward + o.grad_logq_forward,
                            self.g + o.g, self.n + o.n)


def optimiser_code(opt: PolicyGradient):
    """(amc_optimiser id, hyper0, hyper1) of an optimiser object, for the device-resident update."""
    if isinstance(opt, Static):
        return 0, 0.0, 0.0
    if isinstance(opt, VPG):
        return 1, opt.eta, 0.0
    if isinstance(opt, BLPG):
        return 2, opt.eta, 0.0
    if isinstance(opt, BLAPG):
        return 3, opt.delta, opt.eps_id
    if isinstance(opt, NPG):
        return 4, opt.eta, opt.eps_id
    if isinstance(opt, ANPG):
        return 5, opt.delta, opt.eps_id
    if isinstance(opt, BLANPG):
        return 6, opt.delta, opt.eps_id
    raise TypeError(f"No learning_step! is defined for {type(opt).__name__}")


def _sigma_of(parameters) -> float:
    if isinstance(parameters, dict):
        return float(parameters["sigma"])
    return float(np.atleast_1d(np.asarray(parameters, dtype=np.float64))[0])


def log_proposal_density(action, policy, parameters, system=None, device: int = 0) -> float:
    """log_proposal_density(action, policy, parameters, system) for the particle_1d model (particle_1d.jl:52-54),
    evaluated ON THE DEVICE by the estimator's own code (amc_selftest_math, fn 9): there is no host arithmetic for the
    path in this package.  StandardGaussian only (a ScaledGaussian's density needs the system's position)."""
    if type(policy).__name__ != "StandardGaussian":
        raise NotImplementedError("No log_proposal_density is defined for this policy on the host side")      # metropolis.jl:62
    from . import _capi
    return float(_capi.selftest_math("log_proposal_density", [float(action.delta)], [_sigma_of(parameters)], device)[0])


def withgrad_log_proposal_density(grad_out, action, policy, parameters, system=None, device: int = 0) -> float:
    """withgrad_log_proposal_density!(∇logq, action, policy, parameters, system, backend) (gradients.jl:28-33): returns
    logq and writes d logq / d sigma into grad_out[0] -- the value ForwardDiff / Zygote / Enzyme all produce
    (test/ad_backends_test.jl:31-32), computed by the device code the estimator kernel runs."""
    if type(policy).__name__ != "StandardGaussian":
        raise NotImplementedError("No log_proposal_density is defined for this policy on the host side")
    from . import _capi
    d, s = [float(action.delta)], [_sigma_of(parameters)]
    grad_out[0] = float(_capi.selftest_math("grad_log_proposal_density", d, s, device)[0])
    return float(_capi.selftest_math("log_proposal_density", d, s, device)[0])


def initialise_gradient_data(parameters: np.ndarray) -> GradientData:            # gradients.jl:54-61
    z = np.zeros_like(np.asarray(parameters, dtype=np.float64))
    return GradientData(0.0, z.copy(), z.copy(), np.outer(z, z), 0)


def gradient_data_from_row(row: np.ndarray, n_params: int) -> GradientData:
    """One move's row of the engine's estimator output, [j, grad j [P], grad logq [P], g [P][P], n] (AMC_GD_STRIDE_P)."""
    P = int(n_params)
    return GradientData(float(row[0]), np.array(row[1:1 + P]), np.array(row[1 + P:1 + 2 * P]),
                        np.array(row[1 + 2 * P:1 + 2 * P + P * P]).reshape(P, P), int(round(row[1 + 2 * P + P * P])))


def average(gd: GradientData) -> GradientData:                                   # gradients.jl:83-85
    with np.errstate(divide="ignore", invalid="ignore"):
        return GradientData(gd.j / gd.n if gd.n else float("nan"), gd.grad_j / gd.n, gd.grad_logq_forward / gd.n,
                            gd.g / gd.n, gd.n)


def learning_step(parameters: np.ndarray, gd: GradientData, opt: PolicyGradient) -> None:
    """learning_step!(parameters, gd, opt): in-place update, one method per optimiser
    (learning.jl:32-34, 50-52, 77-79, 103-105, 130-134, 160-164)."""
    eye = np.eye(parameters.shape[0])
    if isinstance(opt, VPG):
        parameters[...] = parameters + opt.eta * gd.grad_j
    elif isinstance(opt, BLPG):
        parameters[...] = parameters + opt.eta * (gd.grad_j - gd.j * gd.grad_logq_forward)
    elif isinstance(opt, BLAPG):
        eta = np.sqrt(2 * opt.delta / (np.dot(gd.grad_j, gd.grad_j) + opt.eps_id))
        parameters[...] = parameters + eta * (gd.grad_j - gd.j * gd.grad_logq_forward)
    elif isinstance(opt, NPG):
        parameters[...] = parameters + opt.eta * np.linalg.inv(gd.g + opt.eps_id * eye) @ gd.grad_j
    elif isinstance(opt, ANPG):
        f_inv = np.linalg.inv(gd.g + opt.eps_id * eye)
        eta = np.sqrt(2 * opt.delta / (gd.grad_j @ (f_inv @ gd.grad_j)))
        parameters[...] = parameters + eta * f_inv @ gd.grad_j
    elif isinstance(opt, BLANPG):
        f_inv = np.linalg.inv(gd.g + opt.eps_id * eye)
        bj = gd.grad_j - gd.j * gd.grad_logq_forward
        eta = np.sqrt(2 * opt.delta / (bj @ (f_inv @ bj)))
        parameters[...] = parameters + eta * f_inv @ bj
    elif isinstance(opt, Static):
        pass
    else:
        raise TypeError(f"No learning_step! is defined for {type(opt).__name__}")


# ---- estimator.jl ------------------------------------------------------------------------------
class PolicyGradientEstimator(AriannaAlgorithm):
    """PolicyGradientEstimator(chains; dependencies=(Metropolis,), optimisers, q_batch_size=1, ...)
    (estimator.jl:103-109).  ``ad_backend`` is accepted and ignored: d logq / d sigma of the Gaussian
    policy is evaluated in closed form by the kernel (test/ad_backends_test.jl pins all backends equal)."""

    mutates_chains = True       # every sample leaves x at (x + delta) - delta (gradients.jl:98,103)
    pgmc_role = "estimator"     # run() may issue [Metropolis, this, PolicyGradientUpdate] as one engine call

    def __init__(self, chains, dependencies=None, optimisers=None, q_batch_size: int = 1, ad_backend=None,
                 R=None, parallel: bool = False, device_resident: Optional[bool] = None, **extras):
        assert dependencies is not None and len(dependencies) == 1                 # :104
        assert isinstance(dependencies[0], Metropolis)                             # :105
        self.metropolis: Metropolis = dependencies[0]
        self.pool = self.metropolis.pool
        self.seed = self.metropolis.seed
        self.optimisers = tuple(optimisers)
        assert len(self.optimisers) == len(self.pool)                              # :70
        self.learn_ids: List[int] = [k for k, o in enumerate(self.optimisers) if not isinstance(o, Static)]  # :72
        self.q_batch_size = int(q_batch_size)
        self.parameters_list = [m.parameters for m in self.pool]
        self.objectives = np.zeros(len(self.learn_ids))                            # :83
        self.gradients_data: List[GradientData] = [initialise_gradient_data(self.parameters_list[k])
                                                   for k in self.learn_ids]        # :84
        self.parallel = parallel
        # device_resident: keep gradients_data and the learning step on the GPU (amc_pg_accumulate / amc_pg_update):
        # no host round trip per step.  Default: on when this process holds the whole ensemble (the cross-process
        # sum below goes through torch.distributed on the host); the values are identical either way.
        eng = self.metropolis.engine
        if device_resident is None:
            if self.metropolis.world_size == 1:
                device_resident = hasattr(eng, "pg_accumulate")
            else:
                device_resident = hasattr(eng, "pg_accumulate") and self.connect_shards()
        self.device_resident = bool(device_resident)

    def connect_shards(self) -> bool:
        """Sharded runs: give the engines an RCCL communicator of their own (amc_comm_init), so that the estimator's
        fold is ONE in-place all-reduce of 4 n_learn doubles on each engine's main stream and gradients_data / the learning
        step stay on the devices -- no host round trip per step.  The 128-byte ncclUniqueId travels over the process
        group the script already has.  libamc.so resolves RCCL with dlopen by soname, i.e. it shares the instance
        torch.distributed has loaded.  False (host path via pg_estimate + torch all-reduce) when there is no NCCL
        process group or the engine cannot do it."""
        met = self.metropolis
        if getattr(met, "_comm_connected", False):
            return True
        met._comm_connected = sharding.connect_engine(met.engine)
        return met._comm_connected

    def refresh(self) -> None:
        """Device-resident mode: pull the running gradients_data / objectives to the host (synchronises)."""
        if self.device_resident and self.learn_ids:
            acc = self.metropolis.engine.pg_get_accumulated(self.learn_ids)
            for k, lid in enumerate(self.learn_ids):
                self.gradients_data[k] = gradient_data_from_row(acc[k], self.parameters_list[lid].shape[0])
                self.objectives[k] = acc[k, 0] / acc[k, -1] if acc[k, -1] else 0.0

    def make_step(self, simulation: Simulation) -> None:
        """estimator.jl:111-134: fold GradientData over chains x q_batch samples per learnable move."""
        if not self.learn_ids:
            return
        if self.device_resident:
            self.metropolis.engine.pg_accumulate(self.learn_ids, self.q_batch_size)
            self.metropolis.invalidate_reductions()
            return
        from ._capi import xsum_round
        local = self.metropolis.engine.pg_estimate_exact(self.learn_ids, self.q_batch_size)
        self.metropolis.invalidate_reductions()      # every sample leaves x at (x+d)-d (gradients.jl:103)
        # the `+` fold over the shards: exact integer records merged, rounded once (reproducible sums) -- the same bits for
        # every number of shards
        merged = sharding.allreduce_xsum(local, self.metropolis.engine)
        total = xsum_round(merged).reshape(local.shape[0], local.shape[1])
        for k, lid in enumerate(self.learn_ids):
            gd = gradient_data_from_row(total[k], self.parameters_list[lid].shape[0])
            self.gradients_data[k] = self.gradients_data[k] + gd                   # :130
            self.objectives[k] = self.gradients_data[k].j / self.gradients_data[k].n   # :131

    def make_steps_grouped(self, simulation: Simulation, n: int, update: Optional["PolicyGradientUpdate"],
                           with_reductions: bool = False) -> bool:
        """n time steps that schedule exactly [Metropolis, this estimator(, update)] (PGMC_harmonic_oscillator.jl:24-33
        has them every t) as ONE engine call: same launches in the same order, without ~6 host calls per step.
        False when this configuration cannot be grouped (the caller then steps the algorithms one by one)."""
        eng = self.metropolis.engine
        if not (self.device_resident and self.learn_ids and hasattr(eng, "pgmc_steps")):
            return False
        met = self.metropolis
        met._drop_pending_reduction()
        # with_reductions: a callback observes the state these steps leave -- the sums ride in the last step's launch
        red = bool(with_reductions) and hasattr(eng, "reduce_end_exact")
        if update is not None:
            codes = [optimiser_code(self.optimisers[lid]) for lid in self.learn_ids]
            opt = ([c[0] for c in codes], [c[1] for c in codes], [c[2] for c in codes])
            met.device_params_dirty = True
        else:
            opt = ()
        if red:
            # One reduction in flight per engine: the previous callback's sums are fetched BEFORE the launch that forms the
            # next ones -- but AFTER the n - 1 steps in front of it have been queued, so that the device works through them
            # while the host reads (fetching first left the queue empty for a few microseconds every callback period).
            if n > 1:
                eng.pgmc_steps(n - 1, self.learn_ids, self.q_batch_size, *opt)
            met._settle_claimed(keep=1)          # two reductions in flight per engine (Metropolis._inflight)
            eng.pgmc_steps(1, self.learn_ids, self.q_batch_size, *opt, reduce_begin=True)
        else:
            eng.pgmc_steps(n, self.learn_ids, self.q_batch_size, *opt)
        met._epoch += 1
        met.invalidate_reductions()
        if red:
            met._pending_red_epoch = met._epoch
        return True

    def write_algorithm(self, io, scheduler) -> None:                              # :136-147
        io.write("\tPolicyGradientEstimator\n")
        io.write(f"\t\tCalls: {_calls(scheduler)}\n")
        io.write(f"\t\tLearnable moves: {[k + 1 for k in self.learn_ids]}\n")
        io.write(f"\t\tQ batch size: {self.q_batch_size}\n\t\tAD backend: closed form (HIP kernel)\n")
        io.write(f"\t\tSeed: {self.seed}\n")


# ---- update.jl ------------------------------------------------------------------------------------
class PolicyGradientUpdate(AriannaAlgorithm):
    """PolicyGradientUpdate(chains; dependencies=(PolicyGradientEstimator,)) (update.jl:43-48)."""

    pgmc_role = "update"

    def __init__(self, chains, dependencies=None, **extras):
        assert dependencies is not None and len(dependencies) == 1
        assert isinstance(dependencies[0], PolicyGradientEstimator)
        self.estimator: PolicyGradientEstimator = dependencies[0]
        self.optimisers = self.estimator.optimisers
        self.learn_ids = self.estimator.learn_ids
        self.parameters_list = self.estimator.parameters_list

    def make_step(self, simulation: Simulation) -> None:
        """update.jl:50-57: average -> learning_step! -> reset; then refresh the device copy of sigma."""
        est = self.estimator
        if est.device_resident:
            codes = [optimiser_code(self.optimisers[lid]) for lid in self.learn_ids]
            est.metropolis.engine.pg_update(self.learn_ids, [c[0] for c in codes], [c[1] for c in codes],
                                            [c[2] for c in codes])
            est.metropolis.device_params_dirty = True      # host copies of Move.parameters are refreshed lazily
            return
        for k, lid in enumerate(self.learn_ids):
            gd = average(est.gradients_data[k])
            learning_step(self.parameters_list[lid], gd, self.optimisers[lid])
            est.gradients_data[k] = initialise_gradient_data(self.parameters_list[lid])
            est.metropolis.set_parameters(lid, self.parameters_list[lid])

    def write_algorithm(self, io, scheduler) -> None:                              # update.jl:59-67
        io.write("\tPolicyGradientUpdate\n")
        io.write(f"\t\tCalls: {_calls(scheduler)}\n")
        io.write(f"\t\tLearnable moves: {[k + 1 for k in self.learn_ids]}\n\t\tOptimisers:\n")
        for k, opt in enumerate(self.optimisers, start=1):
            io.write(f"\t\t\tMove {k}: {opt}\n")
