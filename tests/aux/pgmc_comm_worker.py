"""Worker of tests/test_gpu_fullsize.py::test_sharded_pgmc_device_resident_over_rccl: a PGMC run inside a torch.distributed
(NCCL = RCCL) process group, with the estimator's fold all-reduced on the device through the communicator
PolicyGradientEstimator.connect_shards() sets up, against the same run on the host path."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
USE_STORE = os.environ.get("AMC_TEST_GROUP") == "store"        # ranks joined over the plain-socket store only: no torch here
if not USE_STORE:
    import torch                  # first: libamc.so then shares torch's HIP runtime and RCCL (same sonames)
    import torch.distributed as dist
import montecarlo_amd as ma

if USE_STORE:
    grp = ma.sharding.init_store_group()
    WORLD, RANK = grp.world_size, grp.rank
    assert "torch" not in sys.modules
else:
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    WORLD, RANK = dist.get_world_size(), dist.get_rank()
out = {}
for mode in ("comm", "host"):
    chains = ma.ParticleChains.uniform(60_000, 2.0, -2.0, 2.0)
    pool = (ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.2}, 0.6),
            ma.Move(ma.Displacement(0.0), ma.StandardGaussian(), {"sigma": 0.1}, 0.4))
    optimisers = (ma.Static(), ma.VPG(0.3))
    if os.environ.get("AMC_TEST_POLICY") == "drift2":
        # a policy with TWO parameters (delta = theta0 + theta1 z) and a natural-gradient optimiser: P x P metric, records of
        # 1 + 2P + P(P+1)/2 columns per learnable move through the communicator
        pol = ma.ScriptPolicy("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
                              ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"],
                              n_params=2)
        pool = (ma.Move(ma.Displacement(0.0), pol, [0.0, 0.2], 0.6), ma.Move(ma.Displacement(0.0), pol, [0.05, 0.1], 0.4))
        optimisers = (ma.Static(), ma.BLANPG(1e-4, 1e-6))
    # AMC_TEST_DEVICE: every rank on that device (the shared-memory RCCL stand-in of tests/aux/fake_rccl.c lets ranks share a GPU)
    dev = {} if "AMC_TEST_DEVICE" not in os.environ else {"device": int(os.environ["AMC_TEST_DEVICE"])}
    al = (dict(algorithm=ma.Metropolis, pool=pool, seed=42, **dev),
          dict(algorithm=ma.PolicyGradientEstimator, dependencies=(ma.Metropolis,), optimisers=optimisers,
               q_batch_size=2, device_resident=(None if mode == "comm" else False)),
          dict(algorithm=ma.PolicyGradientUpdate, dependencies=(ma.PolicyGradientEstimator,)),
          # callbacks between the grouped time steps: their sums are all-reduced on the engine's communication stream
          # (amc_allreduce_sum) while the estimator's all-reduces sit on its main stream -- one communicator, two streams
          dict(algorithm=ma.StoreCallbacks, callbacks=(ma.callback_energy, ma.callback_acceptance),
               scheduler=ma.build_schedule(120, 10, 10)))
    with tempfile.TemporaryDirectory() as d:
        sim = ma.Simulation(chains, al, 120, path=d, verbose=False)
        est = sim.algorithms[1]
        if mode == "comm" and WORLD == 1:
            # one rank: the automatic choice is the plain device-resident path; take the communicator route explicitly
            assert est.connect_shards()
            est.device_resident = True
        ma.run(sim)
    eng = sim.algorithms[0].engine
    out[mode] = dict(sigma=[m.sigma for m in pool], parameters=[[float(v).hex() for v in m.parameters] for m in pool],
                     device_resident=est.device_resident,
                     connected=bool(getattr(sim.algorithms[0], "_comm_connected", False)), x0=float(chains.x[0]),
                     energy=[[t, float(v)] for t, v in sim.algorithms[3].rows[0]],
                     acceptance=[[t, [float(a) for a in v]] for t, v in sim.algorithms[3].rows[1]],
                     comm=(eng.comm_info() if hasattr(eng, "comm_info") else None), world=WORLD,
                     x_head=[float(v).hex() for v in chains.x[:8]], shard=list(sim.algorithms[0].shard),
                     torch_imported="torch" in sys.modules, hip_runtime=ma._capi.runtime_info()["hip_runtime"])
if RANK == 0:
    print(json.dumps(out))
if USE_STORE:
    ma.sharding.barrier()
else:
    dist.destroy_process_group()
