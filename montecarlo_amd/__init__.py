"""montecarlo_amd: MI355X-native many-chain Metropolis engine behind Arianna.jl's plugin API.

The compute path is libamc.so (hand-written HIP for gfx950, C ABI in include/amc.h).  This
package is the host-side mirror of the reference's interface for that one path -- Move / pool /
Simulation / run -- so a user of TheDisorderedOrganization/MonteCarlo's particle_1d examples finds
the same names.  There is no CPU implementation here: without the built extension and a GPU,
constructing ``Metropolis`` raises.
"""
from ._capi import AmcError, HipEngine, SplitEngine, device_count
from .metropolis import Metropolis, callback_acceptance, callback_energy, callback_moments
from .policy_guided import (ANPG, BLANPG, BLAPG, BLPG, NPG, VPG, GradientData, PolicyGradientEstimator,
                            PolicyGradientUpdate, Static, average, initialise_gradient_data, learning_step,
                            log_proposal_density, withgrad_log_proposal_density)
from . import sharding
from .sharding import allreduce_sum, init_store_group, shard_range
from .simulation import (AriannaAlgorithm, PrintTimeSteps, Simulation, StoreCallbacks, StoreParameters,
                         build_schedule, julia_repr, run)
from .storage import StoreHistogram, StoreSnapshots, checkpoint, restore
from .trajectories import DAT, TXT, StoreBackups, StoreLastFrames, StoreTrajectories
from .system import Action, AriannaSystem, Policy
from .system import CustomPotential, Displacement, Move, ParticleChains, ScaledGaussian, ScriptAction, ScriptPolicy, StandardGaussian, potential

__all__ = [
    "AmcError", "HipEngine", "SplitEngine", "device_count",
    "Metropolis", "callback_acceptance", "callback_energy", "callback_moments",
    "ANPG", "BLANPG", "BLAPG", "BLPG", "NPG", "VPG", "Static", "GradientData", "PolicyGradientEstimator",
    "PolicyGradientUpdate", "average", "initialise_gradient_data", "learning_step",
    "log_proposal_density", "withgrad_log_proposal_density",
    "allreduce_sum", "init_store_group", "shard_range", "sharding",
    "AriannaAlgorithm", "PrintTimeSteps", "Simulation", "StoreCallbacks", "StoreParameters",
    "build_schedule", "julia_repr", "run",
    "StoreHistogram", "StoreSnapshots", "checkpoint", "restore",
    "DAT", "TXT", "StoreBackups", "StoreLastFrames", "StoreTrajectories",
    "Action", "AriannaSystem", "Policy", "CustomPotential", "Displacement", "Move", "ParticleChains", "ScaledGaussian", "ScriptAction", "ScriptPolicy", "StandardGaussian", "potential",
]
