"""GradientData columns nobody reads are not summed -- where nobody CAN read them.

A fused PGMC time step of a script-defined model (amc_pgmc_steps with an update: sweep + estimator + learning step in one launch)
consumes gradients_data in its own tail and resets it (update.jl:50-57), so a column group the move's optimiser never reads --
grad logq_forward for VPG / NPG / ANPG, the metric g for VPG / BLPG / BLAPG (learning.jl:32-164) -- cannot be seen by anybody;
the launch does not deposit it (amc_estimator.h pg_skip_of; -4 ... -8 % per time step).  Everything observable must be the same
bits as with AMC_NO_COLUMN_SKIP=1, for every optimiser, and estimator steps WITHOUT an update in their launch keep every column."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BETA = 2.0
MALA = ("-2.0*sigma*sigma*x + sigma*z",
        "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)",
        "((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(sigma*sigma*sigma) - 4.0*x*(delta + 2.0*sigma*sigma*x)/sigma - 1.0/sigma")
DRIFT = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
         ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"])
KINDS = {"VPG": (1, 0.02, 0.0), "BLPG": (2, 0.02, 0.0), "BLAPG": (3, 1e-4, 1e-6), "NPG": (4, 1e-3, 1e-6), "ANPG": (5, 1e-5, 1e-6),
         "BLANPG": (6, 1e-5, 1e-6)}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _engine(gpu, skip, **kw):
    os.environ["AMC_NO_COLUMN_SKIP"] = "0" if skip else "1"
    try:
        return gpu.HipEngine(**kw)
    finally:
        del os.environ["AMC_NO_COLUMN_SKIP"]


@pytest.mark.parametrize("opt", list(KINDS))
def test_every_optimiser_learns_the_same_bits_with_and_without_the_unread_columns(gpu, oracle, opt):
    kind, h0, h1 = KINDS[opt]
    M = 6001
    out = []
    kw1 = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.5, 0.3], weight=[0.5, 0.5], seed=9, proposal=MALA)
    kw2 = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[[0.05, 0.5]], weight=[1.0], seed=9, proposal=DRIFT, n_params=2)
    for skip in (True, False):
        a, b = _engine(gpu, skip, **kw1), _engine(gpu, skip, **kw2)
        a.init_uniform(-2.0, 2.0)
        b.init_uniform(-2.0, 2.0)
        a.pgmc_steps(5, [0, 1], 2, [kind, 1], [h0, 0.02], [h1, 0.0])          # move 0: the optimiser under test; move 1: VPG
        b.pgmc_steps(5, [0], 2, [kind], [h0], [h1])
        out.append(([a.get_parameters(k)[0] for k in range(2)], a.download_state()[0], a.download_counters(), a.pg_get_accumulated([0, 1]),
                    b.get_parameters(0), b.download_state()[0], b.pg_get_accumulated([0])))
        a.close()
        b.close()
    s, n = out
    assert s[0] == n[0] and np.array_equal(bits(s[1]), bits(n[1])) and np.array_equal(s[2][0], n[2][0]) and np.array_equal(s[2][1], n[2][1])
    assert np.array_equal(s[4], n[4]) and np.array_equal(bits(s[5]), bits(n[5]))
    assert not np.any(s[3]) and not np.any(n[3]) and not np.any(s[6]) and not np.any(n[6])       # consumed and reset either way
    assert s[0][0] != 0.5 and not np.array_equal(s[4], [0.05, 0.5])                              # ... and something was learnt
    # against the oracle, which sums every column: the one-parameter pool
    o = oracle.OracleEngine(**kw1)
    o.init_uniform(-2.0, 2.0)
    for _ in range(5):
        o.sweep(1)
        o.pg_accumulate([0, 1], 2)
        o.pg_update([0, 1], [kind, 1], [h0, 0.02], [h1, 0.0])
    assert s[0] == [o.get_parameters(k)[0] for k in range(2)] and np.array_equal(bits(s[1]), bits(o.download_state()[0]))
    oracle.install_custom_proposal(None)


def test_estimator_steps_without_an_update_keep_every_column(gpu, oracle):
    """PolicyGradientUpdate scheduled less often than the estimator (test/pgmc_test.jl: every 2 steps): the launches that only
    accumulate sum every column -- gradients_data is state somebody may read (amc_pg_get_accumulated, checkpoints) -- and equal the
    oracle's; the launch that updates consumes whatever is there."""
    M = 4099
    kw = dict(n_chains=M, potential="harmonic", beta=BETA, sigma=[0.5], weight=[1.0], seed=3, proposal=MALA)
    e, o = gpu.HipEngine(**kw), oracle.OracleEngine(**kw)
    for x in (e, o):
        x.init_uniform(-2.0, 2.0)
    e.pgmc_steps(3, [0], 2)                          # sweep + estimator, no update: three steps' GradientData on the books
    for _ in range(3):
        o.sweep(1)
        o.pg_accumulate([0], 2)
    acc = e.pg_get_accumulated([0])
    assert np.array_equal(acc, o.pg_get_accumulated([0])) and np.all(acc[0, :4] != 0.0)
    e.pgmc_steps(1, [0], 2, [1], [0.02], [0.0])      # VPG: this launch skips what VPG does not read; the step is the oracle's
    o.sweep(1)
    o.pg_accumulate([0], 2)
    o.pg_update([0], [1], [0.02], [0.0])
    assert e.get_parameters(0)[0] == o.get_parameters(0)[0] and not np.any(e.pg_get_accumulated([0]))
    assert np.array_equal(bits(e.download_state()[0]), bits(o.download_state()[0]))
    e.close()
    oracle.install_custom_proposal(None)
