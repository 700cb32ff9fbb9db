"""Chain sharding across GPUs and the only two cross-shard exchanges of the path.

Chains are independent (src/metropolis.jl:303-307 maps mc_sweep! over chains with no shared
mutable state), so rank r of W owns a contiguous range of GLOBAL chain ids and sweeps need no
collective.  Information crosses chains in exactly two places, both tiny sums:
  * the callbacks (callback_energy particle_1d.jl:68-70, callback_acceptance metropolis.jl:319-321)
  * the GradientData `+` fold of the estimator (src/PolicyGuided/estimator.jl:113-129)
These become ONE all-reduce(sum, f64) of a few dozen bytes: torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).  The Philox counter uses the
global chain id, so results do not depend on W (shard invariance is tested).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def world() -> Tuple[int, int]:
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    # A process group can only exist if the caller has imported torch already; never import it from here:
    # torch bundles its own HIP runtime, and loading it AFTER libamc.so has bound the system one puts two
    # HIP runtimes in the process (the C-ABI RCCL path then fails).  Multi-GPU scripts import torch and call
    # init_process_group before building the Simulation, so torch is always first in that case.
    import sys
    if "torch" not in sys.modules:
        return 0, 1
    try:
        import torch.distributed as dist
    except Exception:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_global: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Half-open range [start, stop) of global chain ids owned by ``rank``.

    Boundaries fall on EVEN ids: two adjacent chains share one Philox Box-Muller draw
    (DESIGN.md §3), so a pair never straddles two shards.
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    n_pairs = (n_global + 1) // 2
    base, rem = divmod(n_pairs, world_size)
    p0 = rank * base + min(rank, rem)
    p1 = p0 + base + (1 if rank < rem else 0)
    return min(2 * p0, n_global), min(2 * p1, n_global)


def allreduce_sum(values: np.ndarray) -> np.ndarray:
    """Sum a small f64 vector over all ranks (no-op for a single process)."""
    rank, size = world()
    values = np.ascontiguousarray(values, dtype=np.float64)
    if size == 1:
        return values
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(values.copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def barrier() -> None:
    _, size = world()
    if size > 1:
        import torch.distributed as dist
        dist.barrier()
