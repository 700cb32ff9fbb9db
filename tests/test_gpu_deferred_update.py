"""The learning step a fused PGMC time step leaves to the next launch's prologue.

Inside one amc_pgmc_steps call a time step that updates may stop at the group sums of its GradientData columns and leave
`learning_step!` (src/PolicyGuided/update.jl:50-57, learning.jl:32-164) to the NEXT launch, whose every block adds the group
rows up, rounds once and takes the step itself while its first load is in flight; the parameter table catches up with the
call's last step.  Nothing observable may depend on it: the engine with deferral, the engine without
(AMC_NO_DEFERRED_UPDATE=1) and the free-running oracle agree bit for bit -- parameters after every call, positions, counters,
callback records, accumulators -- whatever the calls are interleaved with."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


POOLS = {"K1": dict(sigma=[0.3], weight=[1.0], ids=[0], counters=False),
         "K1c": dict(sigma=[0.3], weight=[1.0], ids=[0], counters=True),
         "K2": dict(sigma=[0.2, 0.1], weight=[0.6, 0.4], ids=[1], counters=True),
         "K3two": dict(sigma=[0.3, 0.2, 0.1], weight=[0.5, 0.3, 0.2], ids=[0, 2], counters=True)}
OPTS = [(1, 2e-2, 0.0), (3, 1e-5, 1e-6), (4, 5e-3, 1e-6), (6, 1e-5, 1e-6)]          # VPG, BLAPG, NPG, BLANPG


@pytest.mark.parametrize("pool", sorted(POOLS))
@pytest.mark.parametrize("opt", OPTS, ids=["VPG", "BLAPG", "NPG", "BLANPG"])
def test_deferred_learning_steps_change_nothing(gpu, oracle, monkeypatch, pool, opt):
    cfg = POOLS[pool]
    kind, h0, h1 = opt
    ids = cfg["ids"]
    M = 30011
    kw = dict(n_chains=M, potential="double_well" if pool == "K2" else "harmonic", beta=2.0, sigma=cfg["sigma"], weight=cfg["weight"], seed=7,
              per_chain_counters=cfg["counters"])
    eng = gpu.HipEngine(device=0, **kw)
    monkeypatch.setenv("AMC_NO_DEFERRED_UPDATE", "1")
    plain = gpu.HipEngine(device=0, **kw)
    monkeypatch.delenv("AMC_NO_DEFERRED_UPDATE")
    ref = oracle.OracleEngine(**kw)
    all3 = (eng, plain, ref)
    for e in all3:
        e.init_uniform(-2.0, 2.0)
    k, a0, a1 = [kind] * len(ids), [h0] * len(ids), [h1] * len(ids)

    def same(what):
        for lid in range(len(cfg["sigma"])):
            p = [e.get_parameters(lid) for e in all3]
            assert np.array_equal(bits(p[0]), bits(p[2])) and np.array_equal(bits(p[1]), bits(p[2])), (what, lid, p)

    # stretches of every length: 1 (no successor: nothing deferred), 2, 3, 10; the parameters are read after each (the table is current)
    for n in (1, 2, 3, 10, 2):
        for e in all3:
            e.pgmc_steps(n, ids, 1, k, a0, a1)
        same(f"stretch of {n}")
    # ... with the callback sums in the last launch, and q_batch 2
    for e in all3:
        e.pgmc_steps(4, ids, 2, k, a0, a1, reduce_begin=True)
    recs = [e.reduce_end_exact() for e in all3]
    assert np.array_equal(recs[0][0], recs[2][0]) and np.array_equal(recs[1][0], recs[2][0]) and recs[0][1] == recs[2][1]
    same("observed stretch")
    # interleaved with everything that reads or writes what a pending step touches
    for e in all3:
        e.pgmc_steps(3, ids, 1, k, a0, a1)
        e.sweep(2)                                         # plain sweeps propose with the table's sigma
        e.pgmc_steps(2, ids, 1)                            # estimator steps without an update: accumulators fill
        e.pgmc_steps(3, ids, 1, k, a0, a1)                 # ... so the first of these takes its own step (gradients_data is not zero)
        e.set_parameters(ids[0], [0.25])
        e.pgmc_steps(5, ids, 1, k, a0, a1)
        e.pg_accumulate(ids, 1)
        e.pg_update(ids, k, a0, a1)
        e.pgmc_steps(2, ids, 1, k, a0, a1)
    same("interleaved")
    acc = [e.pg_get_accumulated(ids) for e in all3]
    assert np.array_equal(bits(acc[0]), bits(acc[2])) and np.array_equal(bits(acc[1]), bits(acc[2]))
    x = [e.download_state()[0] for e in all3]
    assert np.array_equal(bits(x[0]), bits(x[2])) and np.array_equal(bits(x[1]), bits(x[2]))
    if cfg["counters"]:
        c = [e.download_counters() for e in all3]
        assert np.array_equal(c[0][0], c[2][0]) and np.array_equal(c[0][1], c[2][1])
    assert np.array_equal(eng.counter_totals()[0], plain.counter_totals()[0])
    eng.close()
    plain.close()


def test_a_pending_step_that_cannot_be_taken_sets_the_status(gpu):
    """A learning step that would leave sigma outside [1e-100, 1e100] is not applied and the status flag is raised -- also when
    the step was left to the next launch's prologue."""
    eng = gpu.HipEngine(n_chains=20000, device=0, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=3)
    eng.init_uniform(-2, 2)
    eng.pgmc_steps(4, [1], 1, [1], [-1e6], [0.0])           # VPG with a huge negative eta: every step is refused
    assert eng.get_parameters(1)[0] == 0.1
    with pytest.raises(gpu.AmcError, match="not applied"):
        eng.pg_get_accumulated([1])
    eng.close()
