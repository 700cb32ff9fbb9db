#!/usr/bin/env python3
"""Generate the polynomial constants of the shared arithmetic spec.

The same literal constants are typed into BOTH independent implementations
(oracle/amc_oracle.c and montecarlo_amd/csrc/amc_math.h); this script is the
single source they were produced from.  Output: C hex-float literals.
"""
from decimal import Decimal, getcontext
from fractions import Fraction
import math

getcontext().prec = 80
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899")


def hexf(d):
    f = float(d)            # correctly rounded from the decimal string
    return f"{f.hex()} /* {f!r} */"


def main():
    print("/* exp: Taylor 1/n!, n = 0..13 */")
    for n in range(14):
        print(f"  E{n} = {hexf(Decimal(1) / Decimal(math.factorial(n)))}")
    print("/* sinpi(r) = r*pi + r^3*S1 + ... : S_k = (-1)^k pi^(2k+1)/(2k+1)!, k=1..7 */")
    for k in range(1, 8):
        c = (PI ** (2 * k + 1)) / Decimal(math.factorial(2 * k + 1))
        print(f"  S{k} = {hexf(-c if k % 2 else c)}")
    print("/* cospi(r) = 1 + r^2*C1 + ... : C_k = (-1)^k pi^(2k)/(2k)!, k=1..8 */")
    for k in range(1, 9):
        c = (PI ** (2 * k)) / Decimal(math.factorial(2 * k))
        print(f"  C{k} = {hexf(-c if k % 2 else c)}")
    pi_hi = float(PI)
    pi_lo = float(PI - Decimal(pi_hi))
    print(f"  PI_HI = {pi_hi.hex()} /* {pi_hi!r} */")
    print(f"  PI_LO = {pi_lo.hex()} /* {pi_lo!r} */")
    ln2 = Decimal(2).ln()
    ln2_hi = float.fromhex("0x1.62e42fee00000p-1")
    print(f"  LN2_HI = {ln2_hi.hex()} /* {ln2_hi!r} */")
    print(f"  LN2_LO = {hexf(ln2 - Decimal(ln2_hi))}")
    print(f"  LOG2E = {hexf(Decimal(1) / ln2)}")
    print(f"  TWO_PI = {(2.0 * math.pi).hex()} /* Julia 2π == 2*Float64(pi) */")


if __name__ == "__main__":
    main()
