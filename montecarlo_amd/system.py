"""The particle_1d model objects a driver script builds (example/particle_1d/particle_1d.jl).

Host-side descriptions only -- the arithmetic runs in the HIP kernels.  Names follow the
reference: ``Particle`` (:9-16), ``Displacement`` (:26-28), ``StandardGaussian`` (:48-50),
``Move`` (src/metropolis.jl:140-162).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

POTENTIALS = ("harmonic", "double_well")


@dataclass(frozen=True)
class CustomPotential:
    """A user-defined ``potential(x)`` for the GPU path.

    In the reference the driver script defines ``potential`` as a free Julia function
    (harmonic_oscillator/MC_harmonic_oscillator.jl:4); a closure cannot cross the C ABI, so the body is
    given as ONE C expression in ``x`` (e.g. ``"x*x*x*x - 2.0*x*x + 0.25*x"``) and the HIP kernels are
    compiled for it at run time (``amc_create_custom``, include/amc.h).  ``+ - * /``, ``sqrt``, ``fabs``,
    ``fma``, ``amc_exp``, ``amc_log`` give results that are bit-reproducible on an IEEE host.
    """
    expr: str

    def __str__(self) -> str:
        return f"custom: {self.expr}"


def potential(name, x):
    """potential(x): harmonic_oscillator/MC_harmonic_oscillator.jl:4 (x^2); double well (x^2-1)^2."""
    if isinstance(name, CustomPotential):
        raise ValueError("a CustomPotential is evaluated by the device kernels only (download_state returns e)")
    x = np.asarray(x, dtype=np.float64)
    if name == "harmonic":
        return x * x
    if name == "double_well":
        q = x * x - 1.0
        return q * q
    raise ValueError(f"unknown potential {name!r}; the HIP engine offers {POTENTIALS}")


class AriannaSystem:
    """abstract type AriannaSystem (src/Arianna.jl:22-23): what a chain is an instance of."""


class Action:
    """abstract type Action (src/metropolis.jl:7)."""


class Policy:
    """abstract type Policy (src/metropolis.jl:14)."""


class ParticleChains(AriannaSystem):
    """``chains::Vector{Particle}`` (particle_1d.jl:9-16) as one SoA ensemble description.

    The reference holds M mutable ``Particle(x, beta, e)`` objects; at M = 1e7 that is an
    HBM-resident f64 array owned by the engine.  This object names the GLOBAL ensemble:
    its size, beta (scalar or per chain), potential, and where the initial positions come
    from -- a host array, or ``uniform(lo, hi)`` drawn on device (the example scripts use
    ``4rand(rng) - 2``, MC_harmonic_oscillator.jl:13).
    After ``finalise`` the arrays ``x`` / ``e`` hold this rank's shard again.
    """

    def __init__(self, n_chains: int, beta, potential="harmonic", x: Optional[np.ndarray] = None,
                 init_uniform: Optional[tuple] = None, reward: Optional[str] = None, dtype: str = "f64"):
        if dtype not in ("f64", "f32"):
            raise ValueError(f"dtype must be 'f64' or 'f32', not {dtype!r}")
        # Particle{T} (particle_1d.jl:9): "f32" keeps x, beta, e and the displacement in Float32 where Julia's promotion
        # rules would (policy parameters, proposal density, acceptance stay Float64); host arrays stay float64 and hold
        # Float32 values
        self.dtype = dtype
        if not isinstance(potential, CustomPotential) and potential not in POTENTIALS:
            raise ValueError(f"unknown potential {potential!r}; the HIP engine offers {POTENTIALS} and CustomPotential(expr)")
        self.n_chains = int(n_chains)
        self.potential = potential
        # reward(action, system) of the policy-guided estimator (src/PolicyGuided/gradients.jl:20; particle_1d.jl:42-44
        # defines delta^2): None = delta^2, or a C expression in `delta` and the new position `x`, compiled like a
        # CustomPotential (amc_create_model)
        self.reward = reward
        self.beta_array = None
        if np.ndim(beta) == 0:
            self.beta = float(beta)
        else:
            self.beta_array = np.ascontiguousarray(beta, dtype=np.float64)
            assert self.beta_array.shape == (self.n_chains,)
            self.beta = float(self.beta_array[0])
        self.x = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        if self.x is not None:
            assert self.x.shape == (self.n_chains,)
        self.init_uniform = init_uniform
        self.e = None
        self.shard = (0, self.n_chains)

    @classmethod
    def uniform(cls, n_chains: int, beta, lo: float = -2.0, hi: float = 2.0, potential="harmonic", reward: Optional[str] = None,
                dtype: str = "f64"):
        return cls(n_chains, beta, potential, init_uniform=(float(lo), float(hi)), reward=reward, dtype=dtype)

    def __len__(self) -> int:
        return self.n_chains


@dataclass
class Displacement(Action):
    """Action: shift x by delta (particle_1d.jl:26-28)."""
    delta: float = 0.0


@dataclass
class ScriptAction(Action):
    """A script-defined one-parameter action on the position -- the reference's Action interface
    (src/metropolis.jl:15-119; Displacement's methods are example/particle_1d/particle_1d.jl:30-40) as two C expressions:
      ``perform``  the position after perform_action!(system, action), from ``x`` and ``delta``   (Displacement: "x + delta")
      ``invert``   the parameter of the inverted action, from ``delta`` and the NEW position ``x`` (Displacement: "-delta")
    A rejected step re-applies the inverted action, like perform_action_cached!.  Used with a ScriptPolicy, whose ``logq``
    must be the density of the move in state space (for a scaling x -> x exp(delta) it carries -log|x exp(delta)|)."""
    perform: str = "x + delta"
    invert: str = "-delta"
    delta: float = 0.0


@dataclass(frozen=True)
class StandardGaussian(Policy):
    """Policy: delta ~ Normal(0, sigma) (particle_1d.jl:48-59)."""

    @staticmethod
    def setup_parameters() -> Dict[str, float]:
        return {"sigma": 1.0}


@dataclass(frozen=True)
class ScaledGaussian(Policy):
    """Policy: delta ~ Normal(0, sigma * scale(x)) -- a script-defined policy of the Gaussian-displacement family whose
    width depends on the state (the reference passes `system` to sample_action! / log_proposal_density,
    src/metropolis.jl:177-182).  ``scale`` is one C expression in the current position ``x`` (CustomPotential's
    vocabulary), compiled for the GPU at run time; the forward density is taken at the old state, the backward one at
    the new state, so the proposal ratio enters the acceptance.  All moves of a pool share the scale expression."""
    scale: str = "1.0"

    @staticmethod
    def setup_parameters() -> Dict[str, float]:
        return {"sigma": 1.0}


@dataclass(frozen=True)
class ScriptPolicy(Policy):
    """A policy whose ``sample_action!`` and ``log_proposal_density`` are the script's own (the reference's generic
    functions, src/metropolis.jl:35-62; example/particle_1d/particle_1d.jl:52-59 are the Gaussian displacement's methods),
    each as one C expression compiled for the GPU at run time:
      ``sample``  delta = f(z, x, sigma) from ONE standard normal variate ``z``, the position ``x`` and the parameter ``sigma``
      ``logq``    log q(delta | x, sigma), the log-density of what ``sample`` returns
      ``dlogq``   d logq / d sigma -- OPTIONAL: the reference gets it from its AD backends (gradients.jl:28-33) and so does the
                  engine when it is None (``logq`` evaluated over dual numbers, ForwardDiff's rules: csrc/amc_dual.h)
    e.g. a drifted Gaussian (Langevin) proposal for U = x^2, beta = 2:
      ScriptPolicy("-2.0*sigma*sigma*x + sigma*z", "-(delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x)/(2.0*sigma*sigma) - amc_log(sigma)", ...)
    All moves of a pool share the policy.

    SEVERAL parameters (Move.parameters is an array in the reference, src/metropolis.jl:140-147; grad j, grad logq and the
    P x P metric g of GradientData follow its shape, PolicyGuided/gradients.jl:41-61): ``n_params`` = P <= 4, the expressions
    say theta0 .. theta{P-1} (``sigma`` stays a name of theta0) and ``dlogq`` lists the P partial derivatives (or is None) -- e.g. a
    Gaussian displacement with a learnable drift,
      ScriptPolicy("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
                   ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"], n_params=2)"""
    sample: str = "sigma*z"
    logq: str = "-(delta*delta)/(2.0*sigma*sigma) - amc_log(sigma)"
    dlogq: Optional[object] = None          # one expression, or the list of the n_params partials
    n_params: int = 1

    def __post_init__(self):
        if isinstance(self.dlogq, (list, tuple)):
            object.__setattr__(self, "dlogq", tuple(self.dlogq))        # hashable: Metropolis compares the moves' policies
            if len(self.dlogq) != self.n_params:
                raise ValueError(f"dlogq must list the {self.n_params} partial derivatives of logq")
            if self.n_params == 1:
                object.__setattr__(self, "dlogq", self.dlogq[0])
        elif self.dlogq is not None and self.n_params != 1:
            raise ValueError(f"dlogq must list the {self.n_params} partial derivatives of logq")
        if not 1 <= int(self.n_params) <= 4:
            raise ValueError("n_params must be in [1, 4]")

    @staticmethod
    def setup_parameters() -> Dict[str, float]:
        return {"sigma": 1.0}


@dataclass
class Move:
    """Move(action, policy, parameters, weight) (src/metropolis.jl:140-162).

    ``parameters`` is a 1-element float array (ComponentArray(sigma=...) in the reference),
    shared by every chain (metropolis.jl:252-260).  ``total_calls`` / ``accepted_calls`` are
    pool-wide sums here; per-chain values come from ``Metropolis.download_counters``.
    """
    action: Displacement
    policy: Policy
    parameters: np.ndarray
    weight: float
    total_calls: int = 0
    accepted_calls: int = 0

    def __post_init__(self):
        p = self.parameters
        if isinstance(p, dict):
            p = [p["sigma"]]
        self.parameters = np.atleast_1d(np.asarray(p, dtype=np.float64)).copy()
        n_params = int(getattr(self.policy, "n_params", 1))
        if self.parameters.shape != (n_params,):
            raise ValueError("StandardGaussian has exactly one parameter (sigma)" if n_params == 1
                             else f"this policy has {n_params} parameters")
        if not isinstance(self.action, (Displacement, ScriptAction)) or not isinstance(self.policy, (StandardGaussian, ScaledGaussian, ScriptPolicy)):
            raise TypeError("the HIP engine supports Displacement / ScriptAction actions with a StandardGaussian, ScaledGaussian or ScriptPolicy policy only")
        if isinstance(self.action, ScriptAction) and not isinstance(self.policy, ScriptPolicy):
            raise TypeError("a ScriptAction needs a ScriptPolicy: No log_proposal_density is defined for it otherwise")
        self.weight = float(self.weight)

    @property
    def sigma(self) -> float:
        return float(self.parameters[0])
