"""julia/PhiloxRNG.jl, executed from its source text by tests/julia_subset.py, against the CPU oracle -- bit for bit.

No Julia exists in this image, so the stub for the reference's `R=` hook (src/metropolis.jl:245,263) cannot run in Julia
here.  The interpreter models the part of the language the stub is written in (Julia's precedence table, literal typing,
integer promotion / wrap-around / checked conversion, 1-based indexing, fused multiply-add), so these tests DO catch what a
silent mistake in the stub would look like on a machine that has Julia: a mask one digit short (0xfff is UInt16, the
shift that follows then happens in 16 bits), `&` binding the C way, an off-by-one table index, a wrong word of the draw.
What they cannot show is that Julia itself accepts the file (module / Random dispatch plumbing); DESIGN.md §10 says so.
"""
import struct

import numpy as np
import pytest

from tests import oracle_lib as O
from tests.julia_subset import Interpreter, JInt, JuliaError, load_philox_stub

U32 = lambda v: JInt(int(v), 32, False)
U64 = lambda v: JInt(int(v), 64, False)
I64 = lambda v: JInt(int(v), 64, True)
bits = lambda d: struct.unpack("<Q", struct.pack("<d", d))[0]


@pytest.fixture(scope="module")
def jl():
    return load_philox_stub()


def words(t):
    return tuple(U32(w) for w in t)


# ---- the interpreter itself: Julia rules the stub relies on ------------------------------------------------------------
def run(src, name="f", args=()):
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.jl"), "w", encoding="utf-8").write(src)
        it = Interpreter(d)
        it.load("t.jl")
        return it.invoke(name, list(args))


def test_interpreter_follows_julia_precedence_and_integer_typing():
    # shifts bind tighter than *, & binds like *, | and xor like +, comparisons below all of them
    assert run("f() = 1 + 2 << 3").v == 17
    assert run("f() = 6 & 3 == 2") is True                       # (6 & 3) == 2, NOT 6 & (3 == 2) as in C
    assert run("f() = 1 | 2 * 4").v == 9
    assert run("f() = -2.0^-2") == -0.25
    # literal widths: 0xfff is UInt16, so a shift of it by 24 happens in 16 bits and loses everything
    r = run("f() = 0xfff << 24")
    assert (r.bits, r.signed, r.v) == (16, False, 0)
    r = run("f() = 0x00000fff << 24")
    assert (r.bits, r.v) == (32, 0xFF000000)
    # promotion and wrap-around
    r = run("f(a, b) = a * b", args=(U64(0xD2511F53), U32(0xFFFFFFFF)))
    assert (r.bits, r.signed, r.v) == (64, False, 0xD2511F53 * 0xFFFFFFFF)
    r = run("f(a) = a + 0x9E3779B9", args=(U32(0xF0000000),))
    assert (r.bits, r.v) == (32, (0xF0000000 + 0x9E3779B9) & 0xFFFFFFFF)
    r = run("f(a) = a - 1", args=(U64(0),))                      # UInt64 with Int64 -> UInt64, wraps
    assert (r.bits, r.signed, r.v) == (64, False, 2**64 - 1)
    # x % T truncates, T(x) checks
    assert run("f(a) = a % UInt32", args=(U64(0x1_2345_6789),)).v == 0x23456789
    with pytest.raises(JuliaError, match="InexactError"):
        run("f(a) = UInt32(a)", args=(U64(0x1_2345_6789),))
    with pytest.raises(JuliaError, match="InexactError"):
        run("f(a) = UInt64(a - 5)", args=(I64(3),))
    # 1-based indexing with bounds, Int / Int -> Float64, non-boolean conditions are errors
    assert run("f(v) = v[1] + v[2]", args=((I64(3), I64(4)),)).v == 7
    with pytest.raises(JuliaError, match="BoundsError"):
        run("f(v) = v[0]", args=((I64(3),),))
    assert run("f() = 1 / 7") == 1 / 7
    with pytest.raises(JuliaError, match="non-boolean"):
        run("f(a) = a & 1 ? 1 : 2", args=(U64(3),))


def test_interpreter_fma_is_fused():
    a, b = 1.0 + 2.0**-30, 1.0 - 2.0**-30
    assert run("f(a, b) = fma(a, b, -1.0)", args=(a, b)) == -(2.0**-60)      # a*b rounds to 1.0 when not fused
    assert run("f(a, b) = a * b - 1.0", args=(a, b)) == 0.0


# ---- the stub against the oracle ---------------------------------------------------------------------------------------
def test_philox_known_answers_through_the_julia_source(jl):
    # Random123 kat_vectors, philox4x32 10 rounds (the vectors tests/test_oracle_kat.py pins the oracle with)
    for ctr, key, out in [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
                          ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
                          ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
                           (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]:
        got = jl.invoke("philox4x32_10", [words(ctr), U32(key[0]), U32(key[1])])
        assert tuple(w.v for w in got) == out
        assert all((w.bits, w.signed) == (32, False) for w in got)


def test_draw_words_match_the_oracle_counter_layout(jl):
    rng = np.random.default_rng(5)
    cases = [(0, 0, 0, 0, 1), (2**64 - 1, 2**40 + 7, 2**47 + 123, 4095, 2), (42, 4_999_999, 2**32, 1, 1)]
    cases += [(int(rng.integers(0, 2**63)), int(rng.integers(0, 2**33)), int(rng.integers(0, 2**48)), int(rng.integers(0, 4096)),
               int(rng.integers(0, 3))) for _ in range(40)]
    for seed, pair, t, draw, stream in cases:
        got = jl.invoke("draw_words", [U64(seed), U64(pair), U64(t), I64(draw), I64(stream)])
        assert [w.v for w in got] == O.draw_words(seed, pair, t, draw, stream), (seed, pair, t, draw, stream)


def test_uniform_maps_and_spare_bits_match_the_oracle(jl):
    L = O.load()
    rng = np.random.default_rng(6)
    edge = [0, 1, 0xFFF, 0x1000, 0xFFFFFF, 0x1000000, 0xFFFFFFFF, 0xFFFFF000, 0x80000000]
    vs = [tuple(int(x) for x in rng.integers(0, 2**32, 4)) for _ in range(200)]
    vs += [(a, b, a ^ b, b) for a in edge for b in edge]
    for v in vs:
        for half in (0, 1):
            a12, p12 = O.spare12(v, half)
            ja = jl.invoke("spare_accept12", [words(v), U64(half)])
            jp = jl.invoke("spare_pick12", [words(v), U64(half)])
            assert (ja.v, jp.v) == (a12, p12) and ja.bits == jp.bits == 32
            lo, hi = v[1], v[3]
            assert bits(jl.invoke("uniform_accept", [ja, U32(lo), U32(hi)])) == bits(L.amo_uniform_accept(a12, lo, hi))
            assert bits(jl.invoke("uniform_pick", [jp, U32(lo)])) == bits(L.amo_uniform_pick(p12, lo))
        assert bits(jl.invoke("uniform_co", [U32(v[0]), U32(v[1])])) == bits(L.amo_uniform_co(v[0], v[1]))
        assert bits(jl.invoke("uniform_oc", [U32(v[0]), U32(v[1])])) == bits(L.amo_uniform_oc(v[0], v[1]))
        assert bits(jl.invoke("angle28", [U32(v[3])])) == bits(L.amo_angle28(v[3]))


def test_logbm_and_sincospi_tables_match_the_oracle_bit_for_bit(jl):
    L = O.load()
    rng = np.random.default_rng(7)
    us = list(rng.random(300)) + [2.0**-52, 2.0**-30, 0.5, 0.70710678, 0.75, 1.0 - 2.0**-53, 1.0, 1e-300 ** 0.05]
    for u in us:
        assert bits(jl.invoke("logbm", [float(u)])) == bits(L.amo_logbm(float(u))), u
    ws = list(2.0 - rng.random(300) * 2.0) + [2.0, 2.0**-27, 1.0, 0.5, 1.5, 1.0 + 2.0**-27, 0.0078125]
    for w in ws:
        s, c = jl.invoke("sincospi_tab", [float(w)])
        so, co = O.sincospi(float(w))
        assert (bits(s), bits(c)) == (bits(so), bits(co)), w


def test_box_muller_matches_the_oracle_bit_for_bit(jl):
    rng = np.random.default_rng(8)
    vs = [tuple(int(x) for x in rng.integers(0, 2**32, 4)) for _ in range(400)]
    vs += [(0, 0, 0, 0), (0xFFFFFFFF,) * 4, (0xFFFFF000, 0xFFFFFFFF, 0, 0xFFFFFFF0), (0, 0, 5, 0x10)]
    for v in vs:
        z = jl.invoke("box_muller", [words(v)])
        zo = O.box_muller(v)
        # u = 1 gives radius -0.0 in Julia's sqrt(-2.0 * 0.0) and +0.0 in the spec; `0 + sigma * z` is +0.0 either way
        assert all(bits(a) == bits(b) or (a == 0.0 and b == 0.0) for a, b in zip(z, zo)), v


def make_rng(jl, seed, stream, chain):
    return jl.construct("PhiloxRNG", [I64(seed), I64(stream)], [I64(seed + chain)])     # rngs[c] = R(seed + c - 1)


def test_constructor_takes_the_reference_seed_argument(jl):
    r = make_rng(jl, 42, 1, 7)
    assert (r.fields["chain"].v, r.fields["calls"].v, r.fields["chain"].bits, r.fields["chain"].signed) == (7, 0, 64, False)
    with pytest.raises(JuliaError, match="InexactError"):
        jl.construct("PhiloxRNG", [I64(42), I64(1)], [I64(41)])               # a seed below the pool's seed is not a chain


@pytest.mark.parametrize("potential, sigmas, weights", [("double_well", (0.1, 1.0), (0.5, 0.5)),
                                                         ("harmonic", (0.7,), (1.0,))])
def test_reference_call_order_through_the_rng_reproduces_an_oracle_trajectory(jl, potential, sigmas, weights):
    """mc_step! consumes rand(Categorical) -> randn -> rand (metropolis.jl:206, particle_1d.jl:57, metropolis.jl:184):
    driving the lone-particle step with THOSE three calls of the Julia generator must walk the oracle's chain."""
    seed, M, T, beta, offset = 2024, 6, 12, 2.0, 10
    sim = O.OracleSim(M, chain_offset=offset, potential=potential, beta=beta, sigma=sigmas, weight=weights, seed=seed)
    x0 = np.linspace(-1.0, 1.0, M)
    sim.set_x(x0)
    pot = {"harmonic": 0, "double_well": 1}[potential]
    L = O.load()
    w = np.asarray(weights, dtype=np.float64)
    tag = ("type", "Float64")
    xs, es = x0.copy(), sim.state()[1].copy()
    rngs = [make_rng(jl, seed, 1, offset + c) for c in range(M)]
    for t in range(T):
        for c in range(M):
            u_pick = jl.invoke("rand", [rngs[c], None])
            k = L.amo_categorical(O._dptr(w), len(w), u_pick)
            z = jl.invoke("randn", [rngs[c], tag])
            u = jl.invoke("rand", [rngs[c], None])
            _, xs[c], es[c] = O.mc_step_explicit(pot, beta, sigmas[k], z, u, xs[c], es[c])
    sim.make_steps(T)
    x_or, e_or = sim.state()
    assert np.array_equal(xs, x_or) and np.array_equal(es, e_or)
    assert len(set(np.round(x_or - x0, 12))) > 1                               # the chains did move
    assert all(r.fields["calls"].v == 3 * T for r in rngs)
    # out of order: the schedule says randn next -> rand must refuse
    r = make_rng(jl, seed, 1, 0)
    jl.invoke("rand", [r, None])
    with pytest.raises(JuliaError, match="expects randn"):
        jl.invoke("rand", [r, None])


def test_estimator_stream_serves_draw_n_of_the_estimator_step(jl):
    seed = 9
    for chain in (0, 1, 12345):
        r = make_rng(jl, seed, 2, chain)
        for est in (0, 3):
            jl.invoke("begin_estimator_step!", [r, I64(est)])
            for n in range(5):
                z = jl.invoke("randn", [r, ("type", "Float64")])
                zo = O.box_muller(O.draw_words(seed, chain >> 1, est, n, 2))[chain & 1]
                assert bits(z) == bits(zo)
