"""Differential fuzz of the HIP path against the CPU oracle: random configurations, random sequences of operations.

The parity tests elsewhere pick their shapes by hand (BASELINE configs, boundary sizes, edge values); this one draws them:
pool size 1..11 with arbitrary (normalised) weights -- cumulative weights that are no multiples of 2^-12 open cells of the
pick table --, sigma over four decades, beta over two, both potentials, ragged ensemble sizes, shards that start at an
arbitrary EVEN global chain id far from 0, sweepstep 1..4, Float64 and Float32 state, K = 1 with and without per-chain
counters; then a random walk over {single-step launch, multi-step launch, callback reduction, sweep with the reduction
formed in the launch (read at once or with sweeps queued behind it), estimator call, [Metropolis, estimator, update] steps in
one engine call with and without the callback sums of their last step, parameter update, counter download, counter upload
around the 16-bit mark}.  After every state-observing operation: positions and energies bit for bit, counters equal,
reductions within RED_RTOL.  AMC_FUZZ_CASES (default 40, ~15 s) and AMC_FUZZ_SEED widen or move the sample.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RED_RTOL = 1e-10
N_CASES = int(os.environ.get("AMC_FUZZ_CASES", "40"))
SEED = int(os.environ.get("AMC_FUZZ_SEED", "20260304"))


def bits(a, dtype):
    return np.ascontiguousarray(a, dtype=np.float64).astype(np.float32).view(np.uint32) if dtype == "f32" \
        else np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def draw_case(rng, sizes=(1, 2, 3, 63, 64, 65, 127, 255, 256, 257, 511, 513, 1000, 2049, 4099, 6001)):
    K = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 6, 7, 8, 11]))      # register fold (<= 4), its two-pass form (5..8), generic fold (> 8)
    w = rng.dirichlet(np.ones(K) * rng.choice([0.5, 1.0, 5.0]))
    if rng.random() < 0.3:                                   # weights on the 2^-12 grid: every cell of the pick table closed
        w = np.maximum(1, np.round(w * 4096)) / 4096
        w[-1] += 1.0 - w.sum()
        if w[-1] <= 0:
            w = np.full(K, 1.0 / K)
    w = w / w.sum()
    M = int(rng.choice(sizes))
    return dict(
        n_chains=M,
        chain_offset=int(rng.choice([0, 2, 254, 256, 2 ** 20 + 2, 2 ** 33 + 6, 2 ** 40])),
        potential=str(rng.choice(["harmonic", "double_well"])),
        beta=float(np.exp(rng.uniform(np.log(0.2), np.log(20.0)))),
        sigma=[float(np.exp(rng.uniform(np.log(1e-3), np.log(10.0)))) for _ in range(K)],
        weight=[float(v) for v in w],
        seed=int(rng.integers(0, 2 ** 63)),
        sweepstep=int(rng.choice([1, 1, 1, 2, 4])),
        dtype=str(rng.choice(["f64", "f64", "f32"])),
        per_chain_counters=bool(K > 1 or rng.random() < 0.5),
    )


def check_state(e, o, case, where):
    x, en = e.download_state()
    xo, eo = o.download_state()
    dt = case["dtype"]
    assert np.array_equal(bits(x, dt), bits(xo, dt)), f"{where}: positions differ\n{case}"
    assert np.array_equal(bits(en, dt), bits(eo, dt)), f"{where}: energies differ\n{case}"


def check_counters(e, o, case, where):
    if case["per_chain_counters"]:
        a, t = e.download_counters()
        ao, to = o.download_counters()
        assert np.array_equal(a, ao) and np.array_equal(t, to), f"{where}: per-chain counters differ\n{case}"
    else:
        a, t = e.counter_totals()
        ao, to = o.counter_totals()
        assert np.array_equal(a, ao) and np.array_equal(t, to), f"{where}: counter totals differ\n{case}"


# ensembles that give every block of a full grid (256 CUs x 6..8 blocks x 512 chains) more than one tile, some of them ragged
LARGE = (786_433, 1_000_003, 1_572_865, 2_000_001, 3_145_729)
N_LARGE = int(os.environ.get("AMC_FUZZ_LARGE_CASES", "6"))


@pytest.mark.parametrize("index", range(N_LARGE))
def test_random_configuration_large_ensemble(gpu, oracle, index):
    """The same walk over ensembles of 0.8 - 3.1 million chains: grid-stride loops with several tiles per block, the peeled
    ragged iteration, the multi-block reductions and the estimator's two-level tail all take part."""
    run_case(gpu, oracle, np.random.default_rng([SEED, 10_000 + index]), f"large {index}", LARGE, threads=8, max_ops=6, max_multi=12)


@pytest.mark.parametrize("index", range(N_CASES))
def test_random_configuration_random_operations(gpu, oracle, index):
    run_case(gpu, oracle, np.random.default_rng([SEED, index]), index)


def run_case(gpu, oracle, rng, index, sizes=None, threads=1, max_ops=16, max_multi=40):
    case = draw_case(rng) if sizes is None else draw_case(rng, sizes)
    kw = {k: v for k, v in case.items() if k != "per_chain_counters"}
    e = gpu.HipEngine(per_chain_counters=case["per_chain_counters"], n_chains_global=case["chain_offset"] + case["n_chains"], **kw)
    o = oracle.OracleEngine(**kw)
    o.threads = threads
    if rng.random() < 0.5:
        lo = float(rng.uniform(-3, 0))
        e.init_uniform(lo, -lo)
        o.init_uniform(lo, -lo)
    else:
        x0 = rng.normal(0.0, 1.5, case["n_chains"])
        if case["dtype"] == "f32":
            x0 = x0.astype(np.float32).astype(np.float64)
        e.upload_state(x0)
        o.upload_state(x0)
    check_state(e, o, case, "start")
    K = len(case["sigma"])
    for op_index in range(int(rng.integers(min(6, max_ops - 1), max_ops))):
        op = rng.choice(["single", "single", "multi", "reduce", "sweep_reduce", "estimate", "pgmc", "pgmc_reduce", "sigma", "counters", "recount"])
        where = f"case {index}, operation {op_index} ({op})"
        if op == "single":
            for _ in range(int(rng.integers(1, 6))):
                e.sweep(1)
                o.sweep(1)
        elif op == "multi":
            n = int(rng.integers(2, max_multi))
            e.sweep(n)
            o.sweep(n)
        elif op in ("reduce", "sweep_reduce"):
            if op == "sweep_reduce":                   # make_step! and the callbacks that follow it at the same t: one launch
                e.sweep_reduce_begin(1)
                o.sweep_reduce_begin(1)
                behind = int(rng.integers(0, 3))       # sweeps queued behind the callback before its sums are read
                if behind:
                    e.sweep(behind)
                r, ro = e.reduce_end(), o.reduce_end()
                if behind:
                    o.sweep(behind)
            else:
                r, ro = e.reduce(), o.reduce()
            scale = np.maximum(np.abs(ro), 1.0)
            assert np.all((np.abs(r - ro) <= RED_RTOL * scale * np.sqrt(case["n_chains"])) | (np.isnan(r) & np.isnan(ro))), \
                f"{where}: reduction differs\n{r}\n{ro}\n{case}"
            if op == "reduce":
                continue
        elif op == "pgmc_reduce":
            # the same time steps with a callback at the last one: the sums of the state the group leaves ride in its last launch
            # (amc_pgmc_steps_reduce_begin; K <= 4 and <= 2 learnable moves, the plain passes otherwise)
            learn = sorted(int(v) for v in rng.choice(K, size=int(rng.integers(1, min(K, 3) + 1)), replace=False))
            n, q = int(rng.integers(1, 5)), int(rng.choice([1, 2]))
            vpg = [oracle.OPTIMISERS["VPG"]] * len(learn)
            e.pgmc_steps(n, learn, q, vpg, [0.0] * len(learn), [0.0] * len(learn), reduce_begin=True)
            o.pgmc_steps(n, learn, q, vpg, [0.0] * len(learn), [0.0] * len(learn), reduce_begin=True)
            r, ro = e.reduce_end(), o.reduce_end()
            scale = np.maximum(np.abs(ro), 1.0)
            assert np.all((np.abs(r - ro) <= RED_RTOL * scale * np.sqrt(case["n_chains"])) | (np.isnan(r) & np.isnan(ro))), \
                f"{where}: reduction of the grouped time steps differs\n{r}\n{ro}\n{case}"
            assert np.all(e.pg_get_accumulated(learn) == 0.0) and e.estimator_step == o.estimator_step
        elif op == "pgmc":
            # [Metropolis, estimator, update] per time step in ONE engine call (fused launches where the engine has them); the
            # learning rate is 0, so sigma stays what it is and the chains can still be compared bit for bit afterwards
            learn = sorted(int(v) for v in rng.choice(K, size=int(rng.integers(1, min(K, 2) + 1)), replace=False))
            n, q = int(rng.integers(1, 6)), int(rng.choice([1, 1, 2]))
            vpg = [oracle.OPTIMISERS["VPG"]] * len(learn)
            e.pgmc_steps(n, learn, q, vpg, [0.0] * len(learn), [0.0] * len(learn))
            o.pgmc_steps(n, learn, q, vpg, [0.0] * len(learn), [0.0] * len(learn))
            for k in learn:
                assert e.get_parameters(k)[0] == o.get_parameters(k)[0], f"{where}: sigma moved\n{case}"
            assert np.all(e.pg_get_accumulated(learn) == 0.0) and e.estimator_step == o.estimator_step
        elif op == "estimate":
            learn = sorted(int(v) for v in rng.choice(K, size=int(rng.integers(1, min(K, 3) + 1)), replace=False))
            q = int(rng.choice([1, 1, 2, 5]))
            g, go = e.pg_estimate(learn, q), o.pg_estimate(learn, q)
            scale = np.abs(go).max(axis=0) + 1.0
            # Float32 state: the reference-ordered summands are compared to a few ulp of Float32-derived quantities
            tol = (1e-10 if case["dtype"] == "f64" else 1e-9) * scale * np.sqrt(case["n_chains"] * q)
            assert np.all(np.abs(g - go) <= tol), f"{where}: estimator sums differ\n{g}\n{go}\n{case}"
        elif op == "sigma":
            k = int(rng.integers(0, K))
            s = float(np.exp(rng.uniform(np.log(1e-3), np.log(10.0))))
            e.set_parameters(k, [s])
            o.set_parameters(k, [s])
            continue
        elif op == "counters":
            check_counters(e, o, case, where)
            continue
        elif op == "recount":
            # resume with other counts (amc_upload_counters): step counts on both sides of the 16-bit mark, where K <= 4 handles
            # switch the width of their counter arrays -- at the upload, or in the middle of the steps that follow
            if not case["per_chain_counters"]:
                continue
            steps = int(rng.choice([0, 17, 65_500, 65_530, 65_535, 65_536, 70_000, 2 ** 20]))
            tot = rng.multinomial(steps, case["weight"], size=case["n_chains"]).T.astype(np.int64)
            acc = (tot * rng.uniform(0, 1, tot.shape)).astype(np.int64)
            e.upload_counters(acc, tot)
            o.upload_counters(acc, tot)
            check_counters(e, o, case, where)
            continue
        check_state(e, o, case, where)
    check_state(e, o, case, f"case {index}, end")
    check_counters(e, o, case, f"case {index}, end")
    assert e.step == o.step
    e.close()
    o.close()
