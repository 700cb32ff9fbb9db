"""Big-integer model of the reproducible sums -- TEST INFRASTRUCTURE.

A third, independent statement of the definition that the oracle (oracle/amc_oracle.c, integer arithmetic on bit patterns)
and the engine (montecarlo_amd/csrc/amc_xsum.h + kernels, floating-point accumulators of a fixed binade) both implement:
exact rational arithmetic with Python's ``fractions``, two passes (the top level of a running-top sum is found first, as the
definition states it), no streaming, no bit tricks beyond ``lsb1``.
"""
from __future__ import annotations

import math
import struct
from fractions import Fraction

LEVEL_BITS = 50
LMIN = -20
LMAX = 19
E_RATIO = -50


def bits(v: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", v))[0]


def from_bits(b: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", b))[0]


def lsb1(v: float) -> float:
    """v with the last bit of its significand set."""
    return from_bits(bits(v) | 1)


def rn(x: Fraction) -> int:
    """Round to nearest integer, ties to even (Python's round on a Fraction)."""
    return round(x)


def rn53_scaled(k: int, e: int) -> float:
    """The integer k rounded to 53 significant bits (int -> float is correctly rounded, ties to even), times 2^e."""
    if k == 0:
        return 0.0
    try:
        return math.ldexp(float(k), e)
    except OverflowError:
        return math.copysign(math.inf, k)


def level_of(v: float) -> int:
    if v == 0.0 or abs(v) < 2.0 ** -1022:
        return LMIN
    e = math.frexp(abs(v))[1] - 1              # ilogb
    return max(LMIN, (e + 1) // LEVEL_BITS)    # floor division


def flags_value(vals):
    nan = any(math.isnan(v) for v in vals)
    pinf = any(v == math.inf for v in vals)
    ninf = any(v == -math.inf for v in vals)
    if nan or (pinf and ninf):
        return math.nan
    if pinf:
        return math.inf
    if ninf:
        return -math.inf
    return None


def sum_q(values, e: int):
    """(K, value) of a kind-Q sum of quantum 2^e; NaN as soon as a summand is not finite."""
    if any(not math.isfinite(v) for v in values):
        return None, math.nan
    q = Fraction(2) ** e
    k = sum(rn(Fraction(lsb1(v)) / q) for v in values)
    return k, rn53_scaled(k, e)


def sum_q_product(xs, ys, e: int):
    if any(not math.isfinite(v) for v in list(xs) + list(ys)):
        return None, math.nan
    q = Fraction(2) ** e
    k = sum(rn(Fraction(lsb1(x)) * Fraction(lsb1(y)) / q) for x, y in zip(xs, ys))
    return k, rn53_scaled(k, e)


def sum_r(values):
    """(top, K1, K2, value) of a running-top sum."""
    # a finite summand of magnitude 2^999 or more lies beyond the last level: it counts as an infinity of its sign
    values = [math.copysign(math.inf, v) if math.isfinite(v) and abs(v) >= 2.0 ** 999 else v for v in values]
    special = flags_value(values)
    finite = [v for v in values if math.isfinite(v)]
    top = max([LMIN] + [level_of(v) for v in finite])
    q1 = Fraction(2) ** (LEVEL_BITS * top)
    q2 = Fraction(2) ** (LEVEL_BITS * (top - 1))
    k1 = k2 = 0
    for v in finite:
        v1 = Fraction(lsb1(v))
        a = rn(v1 / q1)
        r = v1 - a * q1
        rf = float(r)
        assert Fraction(rf) == r            # the low part is exactly a double
        k1 += a
        k2 += rn(Fraction(lsb1(rf)) / q2)
    value = special if special is not None else rn53_scaled(k1 * 2 ** LEVEL_BITS + k2, LEVEL_BITS * (top - 1))
    return top, k1, k2, value


def record_ints(rec):
    """(kind, e, flags, k1, k2) decoded from a 12-double record (32-bit limbs, the top one signed)."""
    def limbs(w):
        r = 0
        for i in (3, 2, 1, 0):
            r = r * 2 ** 32 + int(w[i])
        return r
    return int(rec[0]), int(rec[1]), int(rec[2]), limbs(rec[3:7]), limbs(rec[7:11])
