"""Reproducible cross-chain sums on the device: every sum the path forms across chains -- callback
sums inside a sweep launch, inside a fused PGMC time step, as a pass of their own; the acceptance-ratio sums of the step-log
fold and of the wide-pool pass; the GradientData fold in all its forms -- as RECORDS, bit for bit against the oracle's
integer restatement, for every grid, split and shard layout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _pair(gpu, oracle, M, **kw):
    eng = gpu.HipEngine(n_chains=M, device=0, **kw)
    ref = oracle.OracleSim(M, **{k: v for k, v in kw.items() if k not in ("per_chain_counters", "n_chains_global")})
    return eng, ref


def _records_equal(got, want, what=""):
    got = np.asarray(got).reshape(-1, 12)
    want = np.asarray(want).reshape(-1, 12)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    for i in range(got.shape[0]):
        assert np.array_equal(got[i], want[i]), (what, i, got[i], want[i])


@pytest.mark.parametrize("M", [1, 2, 777, 4099, 300001])
@pytest.mark.parametrize("potential,sigma,weight", [("harmonic", [0.1], [1.0]), ("double_well", [0.1, 1.0], [0.5, 0.5]),
                                                    ("harmonic", [0.3, 0.2, 0.1, 0.05], [0.25] * 4)])
def test_callback_records_equal_the_oracles(gpu, oracle, M, potential, sigma, weight):
    kw = dict(potential=potential, beta=2.0, sigma=sigma, weight=weight, seed=11, per_chain_counters=True)
    eng, ref = _pair(gpu, oracle, M, **kw)
    eng.init_uniform(-2.0, 2.0)
    ref.init_uniform(-2.0, 2.0)
    # before the first step: 0/0 = NaN ratios (metropolis.jl:320), sums over the initial state
    rec, steps = eng.reduce_exact()
    _records_equal(rec, ref.callback_records(), "t = 0")
    assert steps == 0
    # sums formed inside the sweep launch (single step, then a fused stretch), ratio sums by the fold of the step log
    for n in (1, 7):
        eng.sweep_reduce_begin(n)
        ref.make_steps(n)
        rec, steps = eng.reduce_end_exact()
        _records_equal(rec, ref.callback_records(), f"fused {n}")
        assert steps == ref.step
    # ... and as a pass of its own
    eng.sweep(3)
    ref.make_steps(3)
    rec, steps = eng.reduce_exact()
    _records_equal(rec, ref.callback_records(), "separate pass")
    out = eng.reduce_records_value(rec, steps)
    assert out[0] / M == ref.energy()
    assert np.array_equal(out[4:] / M, ref.acceptance(), equal_nan=True)      # NaN where a chain never picked the move
    assert np.array_equal(bits(out[1:3]), bits(ref.moments()))
    eng.close()


def test_callback_records_of_a_wide_pool(gpu, oracle):
    """More than four moves: the per-move ratio sums come from the pass over the counters (integer atomics per block)."""
    K = 7
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.05 * (k + 1) for k in range(K)], weight=[0.4] + [0.1] * 6, seed=3)
    for M in (5, 40001):
        eng, ref = _pair(gpu, oracle, M, **kw)
        eng.init_uniform(-1.0, 1.0)
        ref.init_uniform(-1.0, 1.0)
        eng.sweep(25)
        ref.make_steps(25)
        rec, _ = eng.reduce_exact()
        _records_equal(rec, ref.callback_records(), f"K = 7, M = {M}")
        eng.sweep_reduce_begin(2)
        ref.make_steps(2)
        _records_equal(eng.reduce_end_exact()[0], ref.callback_records())
        eng.close()


def test_pool_wide_counter_ratio_record_is_the_accepted_total(gpu, oracle):
    M = 10007
    eng = gpu.HipEngine(n_chains=M, device=0, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=5,
                        per_chain_counters=False)
    ref = oracle.OracleEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=5,
                              per_chain_counters=False)
    eng.init_uniform(-2, 2)
    ref.init_uniform(-2, 2)
    eng.sweep_reduce_begin(9)
    ref.sweep_reduce_begin(9)
    a, sa = eng.reduce_end_exact()
    b, sb = ref.reduce_end_exact()
    _records_equal(a, b)
    assert sa == sb == 9
    assert np.array_equal(bits(eng.reduce_records_value(a, sa)), bits(ref.reduce_records_value(b, sb)))
    eng.close()


def test_callback_sums_with_wild_positions(gpu, oracle):
    """Positions over many binades, huge, tiny, zero: the running top rises inside the launch; infinities and NaN flag."""
    rng = np.random.default_rng(2)
    M = 5000
    x = rng.standard_normal(M) * np.exp2(rng.integers(-80, 80, M).astype(np.float64))
    x[::97] = 0.0
    x[5] = 1e150
    for special in (None, np.inf, -np.inf, np.nan):
        xs = x.copy()
        if special is not None:
            xs[1234] = special
        eng, ref = _pair(gpu, oracle, M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1,
                         per_chain_counters=True)
        eng.upload_state(xs)
        ref.set_x(xs)
        rec, _ = eng.reduce_exact()
        want = ref.callback_records()
        _records_equal(rec[:4], want[:4], f"special {special}")
        eng.close()


@pytest.mark.parametrize("n_parts", [2, 3, 5])
def test_sums_do_not_depend_on_the_split_into_shards(gpu, oracle, n_parts):
    """1 shard vs n shards on one device: the merged records, hence every callback row and every GradientData sum, are
    bit-identical -- and equal to the oracle's over the whole ensemble."""
    M = 60001
    kw = dict(potential="double_well", beta=2.0, sigma=[0.2, 0.9], weight=[0.6, 0.4], seed=21)
    whole = gpu.HipEngine(n_chains=M, device=0, **kw)
    split = gpu.SplitEngine(n_chains=M, n_parts=n_parts, device=0, **kw)
    ref = oracle.OracleSim(M, **kw)
    for e in (whole, split, ref):
        e.init_uniform(-2.0, 2.0)
    for e in (whole, split):
        e.sweep(12)
    ref.make_steps(12)
    a, _ = whole.reduce_exact()
    b, _ = split.reduce_exact()
    _records_equal(a, b)
    _records_equal(a, ref.callback_records())
    ga = whole.pg_estimate_exact([0, 1], 3)
    gb = split.pg_estimate_exact([0, 1], 3)
    _records_equal(ga, gb)
    _records_equal(ga, ref.pg_estimate_records([0, 1], 3))
    assert np.array_equal(bits(whole.pg_estimate([1], 2)), bits(split.pg_estimate([1], 2)))
    whole.close()
    split.close()


@pytest.mark.parametrize("potential", ["harmonic", "double_well"])
@pytest.mark.parametrize("M,q_batch", [(1, 1), (2, 3), (4099, 1), (4099, 10), (100003, 2), (513, 40)])
def test_gradient_data_records_equal_the_oracles(gpu, oracle, potential, M, q_batch):
    K = 3
    kw = dict(potential=potential, beta=2.0, sigma=[0.2, 0.1, 1.3], weight=[0.5, 0.3, 0.2], seed=7)
    eng, ref = _pair(gpu, oracle, M, **kw)
    eng.init_uniform(-2.0, 2.0)
    ref.init_uniform(-2.0, 2.0)
    eng.sweep(5)
    ref.make_steps(5)
    for ids in ([1], [0, 2], [0, 1, 2]):
        got = eng.pg_estimate_exact(ids, q_batch)
        want = ref.pg_estimate_records(ids, q_batch)
        _records_equal(got, want, f"learn ids {ids}")
        assert np.array_equal(bits(eng.download_state()[0]), bits(ref.state()[0]))
    # the rounded sums through the plain entry point
    assert np.array_equal(bits(eng.pg_estimate([2], q_batch)), bits(ref.pg_estimate([2], q_batch)))
    eng.close()


def test_gradient_data_of_many_learnable_moves(gpu, oracle):
    K = 8
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.1 + 0.07 * k for k in range(K)], weight=[0.125] * K, seed=9)
    eng, ref = _pair(gpu, oracle, 20011, **kw)
    eng.init_uniform(-1.5, 1.5)
    ref.init_uniform(-1.5, 1.5)
    ids = list(range(K))
    _records_equal(eng.pg_estimate_exact(ids, 2), ref.pg_estimate_records(ids, 2))
    _records_equal(eng.pg_estimate_exact(ids[:5], 1), ref.pg_estimate_records(ids[:5], 1))
    eng.close()


def test_gradient_data_with_nan_state_is_nan(gpu, oracle):
    M = 3000
    x = np.linspace(-1, 1, M)
    x[77] = np.nan
    eng, ref = _pair(gpu, oracle, M, potential="harmonic", beta=2.0, sigma=[0.5], weight=[1.0], seed=1)
    eng.upload_state(x)
    ref.set_x(x)
    got, want = eng.pg_estimate([0], 1), ref.pg_estimate([0], 1)
    assert np.isnan(got[0, 0]) and np.isnan(got[0, 1]) and np.isnan(want[0, 0])
    assert np.array_equal(bits(got[0, 2:]), bits(want[0, 2:]))        # grad logq and g do not see the state
    eng.close()


@pytest.mark.parametrize("q_batch,stretches", [(1, (1, 3, 50, 146)), (24, (1, 2, 7))])
def test_free_running_pgmc_is_bit_exact_against_the_oracle(gpu, oracle, q_batch, stretches):
    """BASELINE config 5 in small: [Metropolis, estimator, update] time steps on the device against the oracle running
    on its own -- nothing fed back: the learned sigma, the positions and the per-chain counters are EQUAL after every
    stretch, because the GradientData fold is the same integer sum on both sides.
    q_batch 24: more summands per trip (48) than a lane's accumulators take between two flushes -- such a time step is two
    launches, the estimator's in its flushing form (pg_fits_without_flush, amc_api.hip)."""
    M = 100003
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=42)
    eng = gpu.HipEngine(n_chains=M, device=0, **kw)
    ref = oracle.OracleEngine(n_chains=M, **kw)
    eng.init_uniform(-2.0, 2.0)
    ref.init_uniform(-2.0, 2.0)
    ids, kinds, h0, h1 = [1], [1], [0.05], [0.0]          # VPG(0.05) on move 2, Static on move 1
    for stretch in stretches:
        eng.pgmc_steps(stretch, ids, q_batch, kinds, h0, h1, reduce_begin=True)
        ref.pgmc_steps(stretch, ids, q_batch, kinds, h0, h1, reduce_begin=True)
        a, sa = eng.reduce_end_exact()
        b, sb = ref.reduce_end_exact()
        assert eng.get_parameters(1)[0] == ref.get_parameters(1)[0], f"sigma after {stretch} more steps"
        _records_equal(a, b, "callback records")
        assert sa == sb
    assert eng.get_parameters(0)[0] == 0.2
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    ac, tc = eng.download_counters()
    ao, to = ref.download_counters()
    assert np.array_equal(ac, ao) and np.array_equal(tc, to)
    assert 0.1 < eng.get_parameters(1)[0] < 2.0 and eng.get_parameters(1)[0] != 0.1          # it learns
    eng.close()


@pytest.mark.parametrize("opt", [("VPG", 1, 1e-3, 0.0), ("BLPG", 2, 1e-3, 0.0), ("BLAPG", 3, 1e-6, 1e-6), ("NPG", 4, 1e-2, 1e-6),
                                 ("ANPG", 5, 1e-6, 1e-6), ("BLANPG", 6, 1e-6, 1e-6)])
def test_free_running_pgmc_every_optimiser(gpu, oracle, opt):
    """The reference's pgmc_test shape (7 moves, q_batch 10, update every 2 steps) for each optimiser, separate launches and
    the accumulators kept across steps: sigma equal to the oracle's at every update."""
    name, kind, h0, h1 = opt
    M = 1001
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.2] * 7, weight=[0.4] + [0.1] * 6, seed=42)
    eng = gpu.HipEngine(n_chains=M, device=0, **kw)
    ref = oracle.OracleEngine(n_chains=M, **kw)
    eng.init_uniform(-2.0, 2.0)
    ref.init_uniform(-2.0, 2.0)
    ids = [1, 2, 3]
    for t in range(40):
        for e in (eng, ref):
            e.sweep(1)
            e.pg_accumulate(ids, 10)
            if t % 2 == 1:
                e.pg_update(ids, [kind] * 3, [h0] * 3, [h1] * 3)
        if t % 2 == 1:
            for k in ids:
                assert eng.get_parameters(k)[0] == ref.get_parameters(k)[0], (name, t, k)
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    eng.close()


def test_two_reductions_in_flight(gpu, oracle):
    M = 50001
    kw = dict(potential="harmonic", beta=2.0, sigma=[0.1, 0.3], weight=[0.5, 0.5], seed=8)
    eng, ref = _pair(gpu, oracle, M, **kw)
    eng.init_uniform(-2, 2)
    ref.init_uniform(-2, 2)
    eng.sweep_reduce_begin(2)
    ref.make_steps(2)
    first = ref.callback_records()
    eng.sweep(5)
    eng.reduce_begin()                      # a second one while the first has not been fetched
    ref.make_steps(5)
    second = ref.callback_records()
    eng.sweep(3)
    with pytest.raises(gpu.AmcError, match="already in flight"):
        eng.reduce_begin()
    with pytest.raises(gpu.AmcError, match="already in flight"):
        eng.sweep_reduce_begin(1)
    _records_equal(eng.reduce_end_exact()[0], first, "oldest first")
    eng.sweep_reduce_begin(1)               # a ticket is free again
    ref.make_steps(4)
    third = ref.callback_records()
    _records_equal(eng.reduce_end_exact()[0], second)
    _records_equal(eng.reduce_end_exact()[0], third)
    with pytest.raises(gpu.AmcError, match="no reduction in flight"):
        eng.reduce_end()
    eng.close()


@pytest.mark.gpu
def test_wave_totals_by_folding_equal_plain_sums():
    """wave_total_i64 (half-wave / row swaps, then an in-row scan in inline assembly) against numpy's sums modulo 2^64 and
    against the plain DPP form, on random, extreme and carry-provoking lane values."""
    from montecarlo_amd import _capi as A
    rng = np.random.default_rng(5)
    cases = [rng.integers(-2**62, 2**62, size=(6, 64), dtype=np.int64),
             rng.integers(-2**31, 2**31, size=(6, 64), dtype=np.int64),
             np.full((6, 64), 0xFFFFFFFF, dtype=np.int64),                       # every add carries out of the low word
             np.full((6, 64), -1, dtype=np.int64),
             (np.arange(6 * 64, dtype=np.int64).reshape(6, 64) + 1) * 0x100000001,
             np.where(np.arange(64)[None, :] % 2 == 0, np.int64(2**62), np.int64(-2**62)) * np.ones((6, 1), dtype=np.int64)]
    one_hot = np.zeros((6, 64), dtype=np.int64)
    for lane in (0, 15, 16, 31, 32, 47, 48, 63):                               # where each lane ends up: a single non-zero lane per value
        one_hot[:] = 0
        for i in range(6):
            one_hot[i, (lane + 7 * i) % 64] = (i + 1) * (1 << 40) + lane
        cases.append(one_hot.copy())
    for v in cases:
        out, plain = A.selftest_wave_totals(v)
        want = v.astype(np.uint64).sum(axis=1, dtype=np.uint64).astype(np.int64)     # modulo 2^64
        assert np.array_equal(plain, want)
        assert np.array_equal(out[:6], want)
        assert np.array_equal(out[6:8], want[[4, 1]])
        assert np.array_equal(out[8:11], want[[5, 0, 2]])
        assert out[11] == want[3]
        assert out[12] == int((v[0].astype(np.uint64) & np.uint64(0xFFFFFFFF)).max())


@pytest.mark.parametrize("cols", [0, 1, 2, 4, 3, 5, 6, 7])
@pytest.mark.parametrize("potential,sigma,weight", [("harmonic", [0.1], [1.0]), ("double_well", [0.1, 1.0], [0.5, 0.5])])
def test_only_the_sums_asked_for_are_formed(gpu, oracle, cols, potential, sigma, weight):
    """amc_set_reduce_columns: callback_energy needs sum e alone (particle_1d.jl:68-70), the moments sum x and sum x^2.  The
    records of the sums that were asked for are the oracle's, the others stay empty (NaN as values) -- in the sweep launch
    that forms them, in a fused stretch and in the pass of its own."""
    M = 30011
    kw = dict(potential=potential, beta=2.0, sigma=sigma, weight=weight, seed=21, per_chain_counters=True)
    eng, ref = _pair(gpu, oracle, M, **kw)
    eng.init_uniform(-2.0, 2.0)
    ref.init_uniform(-2.0, 2.0)
    eng.set_reduce_columns(cols)

    def check(rec, steps, what):
        want = np.asarray(ref.callback_records()).reshape(-1, 12).copy()
        for c in range(3):
            if not cols & (1 << c):
                want[c] = 0.0
        _records_equal(rec, want, what)
        out = eng.reduce_records_value(rec, steps)
        for c in range(3):
            assert np.isnan(out[c]) == (not cols & (1 << c))
        assert out[3] == M
    for n in (1, 5):
        eng.sweep_reduce_begin(n)
        ref.make_steps(n)
        check(*eng.reduce_end_exact(), f"fused {n}")
    eng.sweep(2)
    ref.make_steps(2)
    check(*eng.reduce_exact(), "separate pass")
    eng.set_reduce_columns(7)
    _records_equal(eng.reduce_exact()[0], ref.callback_records(), "all again")
    with pytest.raises(gpu.AmcError):
        eng.set_reduce_columns(8)
    eng.close()


@pytest.mark.parametrize("cols", [1, 3, 7])
@pytest.mark.parametrize("potential,sigma,weight,counters", [("harmonic", [0.1], [1.0], False), ("harmonic", [0.1], [1.0], True),
                                                             ("double_well", [0.2, 0.1], [0.6, 0.4], True)])
def test_fused_pgmc_steps_form_the_sums_asked_for(gpu, oracle, cols, potential, sigma, weight, counters):
    """The fused PGMC time step that also leaves the callback sums (amc_pgmc_steps_reduce_begin) in both of its forms -- sum e
    alone compiled in (cols = 1; harmonic: also with sum x^2, the same sum), the columns of a run-time mask otherwise -- for the
    three sweep forms it rides on (K = 1 with the pool-wide counter, K = 1 with per-chain counters, K = 2): records of the sums asked
    for equal the oracle's, the others stay empty; sigma and the chains equal a free-running oracle."""
    M = 40009
    kw = dict(potential=potential, beta=2.0, sigma=sigma, weight=weight, seed=33, per_chain_counters=counters)
    eng = gpu.HipEngine(n_chains=M, device=0, **kw)
    ref = oracle.OracleEngine(n_chains=M, **kw)
    lid = len(sigma) - 1
    for e in (eng, ref):
        e.init_uniform(-2.0, 2.0)
        e.set_reduce_columns(cols)
    for n in (1, 4):
        for e in (eng, ref):
            e.pgmc_steps(n, [lid], 1, [1], [1e-3], [0.0], reduce_begin=True)
        rec, steps = eng.reduce_end_exact()
        rec_o, steps_o = ref.reduce_end_exact()
        assert steps == steps_o
        _records_equal(rec, rec_o, f"{n} fused steps, columns {cols}")
        out = eng.reduce_records_value(rec, steps)
        for c in range(3):
            assert np.isnan(out[c]) == (not cols & (1 << c))
    assert eng.get_parameters(lid)[0] == ref.get_parameters(lid)[0]
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    eng.close()


def test_compact_and_wide_block_rows_carry_the_same_sums(gpu, oracle, monkeypatch):
    """A launch whose lanes add at most 32 summands per column leaves ONE 64-byte row per block (tops and flags packed, 64-bit
    totals); beyond that, and with AMC_WIDE_RED_ROWS=1, the wide row.  Same records either way -- plain values, wild ones
    (tops that differ between the waves of a block, NaN and infinities) and a pool-wide counter."""
    rng = np.random.default_rng(8)
    M = 70001
    wild = rng.standard_normal(M) * np.exp2(rng.integers(-120, 120, M).astype(np.float64))
    wild[::977] = np.inf
    wild[5::1999] = np.nan
    for x0, counters in ((None, True), (wild, True), (None, False)):
        recs = []
        for wide in ("0", "1"):
            monkeypatch.setenv("AMC_WIDE_RED_ROWS", wide)
            eng = gpu.HipEngine(n_chains=M, device=0, potential="double_well" if counters else "harmonic", beta=2.0,
                                sigma=[0.1, 1.0] if counters else [0.1], weight=[0.5, 0.5] if counters else [1.0], seed=4,
                                per_chain_counters=counters)
            if x0 is None:
                eng.init_uniform(-2, 2)
            else:
                eng.upload_state(x0)
            eng.sweep_reduce_begin(1)
            a = eng.reduce_end_exact()
            b = eng.reduce_exact()
            recs.append((a, b))
            eng.close()
        for (ra, sa), (rb, sb) in zip(recs[0], recs[1]):
            _records_equal(ra, rb)
            assert sa == sb
    monkeypatch.delenv("AMC_WIDE_RED_ROWS")


def test_more_than_four_learnable_moves_go_in_launches_of_four(gpu, oracle, monkeypatch):
    """The pool of the reference's test/pgmc_test.jl:16-27 -- seven Gaussian displacements, six learnable, one optimiser each,
    q_batch_size = 10 -- where the samples need the estimator's flushing form: the call goes in two launches (moves 1-4, 5-6;
    the draws named by the move's index in the call) instead of one of capacity 8.  Parameters, positions, accumulators: the same
    bits as the one launch (AMC_NP_SMALL_LAUNCHES=1 keeps it) and as the free-running oracle."""
    M = 40001
    kw = dict(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1] * 7, weight=[0.4] + [0.1] * 6, seed=17)
    learn, kinds = [1, 2, 3, 4, 5, 6], [1, 2, 3, 4, 5, 6]
    h0, h1 = [0.001, 0.001, 1e-6, 1e-2, 1e-6, 1e-6], [0.0, 0.0, 1e-6, 1e-6, 1e-6, 1e-6]
    eng = gpu.HipEngine(**kw)
    monkeypatch.setenv("AMC_NP_SMALL_LAUNCHES", "1")
    one = gpu.HipEngine(**kw)
    monkeypatch.delenv("AMC_NP_SMALL_LAUNCHES")
    ref = oracle.OracleEngine(**kw)
    for e in (eng, one, ref):
        e.init_uniform(-2.0, 2.0)
    q = 20                                               # a lane's accumulators take 32 summands between flushes: 2 x 20 per pair calls for the flushing form
    for stretch in (1, 2):
        for e in (eng, one, ref):
            e.pgmc_steps(stretch, learn, q, kinds, h0, h1)
        for k in range(7):
            assert eng.get_parameters(k)[0] == ref.get_parameters(k)[0] == one.get_parameters(k)[0], (stretch, k)
    for e in (eng, one, ref):
        e.sweep(1)
        e.pg_accumulate(learn, q)                        # no update: gradients_data holds the sums
    acc = eng.pg_get_accumulated(learn)
    assert np.array_equal(acc, ref.pg_get_accumulated(learn)) and np.array_equal(acc, one.pg_get_accumulated(learn))
    assert acc[0, 4] == M * q and np.all(acc[:, 0] > 0)
    x = eng.download_state()[0]
    assert np.array_equal(x.view(np.uint64), ref.download_state()[0].view(np.uint64))
    assert np.array_equal(x.view(np.uint64), one.download_state()[0].view(np.uint64))
    assert eng.estimator_step == ref.estimator_step == one.estimator_step
    eng.close()
    one.close()
