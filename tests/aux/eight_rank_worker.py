"""Worker of tests/test_gpu_fake_rccl.py::test_eight_ranks_on_one_gpu: the machine's real shape -- EIGHT engine ranks in one
communicator -- on a box that has one GPU and lets six processes use it.

The engine's communicator belongs to a HANDLE (amc_comm_init(h, rank, n_ranks, id)), not to a process, so one process can hold
several ranks: every process started by the launcher hosts RANKS_PER_PROCESS of them, each on a thread of its own with a handle
and stream of its own (the stand-in's collectives block the calling thread until every rank has arrived, as ncclCommInitRank
does).  With 4 processes x 2 ranks the library runs with comm_ranks = 8: shards keyed by global chain id, the gather payload
[8][sums][12], pg_merge_slots over eight slots, n_samples of the global ensemble, callbacks' sums on the communication stream.
The work is BASELINE config 5's shape (device-resident PGMC, callbacks every 10 time steps), driven through the engine's
own entry points (the host mirror's sharding module is one rank per process).  Output: one JSON line from global rank 0."""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import montecarlo_amd as ma
from montecarlo_amd import _capi as A

RPP = int(os.environ.get("AMC_TEST_RANKS_PER_PROCESS", "2"))
M_GLOBAL = int(os.environ.get("AMC_TEST_CHAINS", "60000"))
PERIODS, PERIOD = 12, 10
grp = ma.sharding.init_store_group()          # the processes meet over the plain-socket store (no torch in the worker)
N_RANKS = grp.world_size * RPP
assert "torch" not in sys.modules

# global rank 0 makes the unique id, the store carries it to every process
uid = grp.broadcast(A.HipEngine.comm_unique_id() if grp.rank == 0 else None, src=0)

results = [None] * RPP
errors = []


def one_rank(slot):
    try:
        rank = grp.rank * RPP + slot
        start, stop = ma.sharding.shard_range(M_GLOBAL, rank, N_RANKS)
        e = A.HipEngine(n_chains=stop - start, chain_offset=start, n_chains_global=M_GLOBAL, device=int(os.environ.get("AMC_TEST_DEVICE", "0")),
                        potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=42)
        e.init_uniform(-2.0, 2.0)
        e.set_reduce_columns(1)
        e.comm_init(rank, N_RANKS, uid)              # returns when all N_RANKS have joined
        rows = []
        for p in range(PERIODS):
            e.pgmc_steps(PERIOD - 1, [1], 2, [1], [0.3], [0.0])
            e.pgmc_steps(1, [1], 2, [1], [0.3], [0.0], reduce_begin=True)
            rec, steps = e.reduce_end_exact()
            red = e.reduce_records_value(e.allreduce_xsum(rec), steps)
            rows.append([float(red[0] / red[3]).hex(), [float(v / red[3]).hex() for v in red[A.AMC_RED_HEADER:]], int(round(red[3]))])
        # one estimator call on the host path too: records over the same communicator
        g = e.allreduce_xsum(e.pg_estimate_exact([1], 2).reshape(-1, A.AMC_XSUM_WORDS))
        info = e.comm_info()
        x = e.download_state()[0]
        results[slot] = dict(rank=rank, shard=[start, stop], sigma=[float(e.get_parameters(k)[0]).hex() for k in range(2)], rows=rows,
                             comm=info, x_head=[float(v).hex() for v in x[:8]], gd=[float(v).hex() for v in A.xsum_round(g)],
                             x_checksum="%016x" % int(np.bitwise_xor.reduce(x.view(np.uint64))))
        e.comm_destroy()
        e.close()
    except BaseException as exc:          # noqa: BLE001 -- a thread's failure must reach the process's exit status
        errors.append(repr(exc))


threads = [threading.Thread(target=one_rank, args=(s,)) for s in range(RPP)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=300)
if errors or any(r is None for r in results):
    print("rank thread failed:", errors, file=sys.stderr)
    sys.exit(3)
everything = grp.allgather(results)
if grp.rank == 0:
    flat = sorted((r for per_process in everything for r in per_process), key=lambda r: r["rank"])
    checksum = 0
    for r in flat:
        checksum ^= int(r["x_checksum"], 16)
    print(json.dumps(dict(n_ranks=N_RANKS, processes=grp.world_size, ranks=flat, x_checksum="%016x" % checksum)))
ma.sharding.barrier()
