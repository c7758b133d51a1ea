#!/usr/bin/env python3
"""The cumulative table of the encryption noise sampler (csrc/encrypt.hip kNoiseCdt, oracle/fhe_oracle.c fo_noise_cdt),
recomputed with 90 decimal digits.  The error polynomials of a fresh encryption are rounded normals with sigma = 3.19,
redrawn beyond 19 (SEAL's default noise_standard_deviation and 6 sigma clip): for |v| <= 19
    P(e = v) = [Phi((v + 1/2) / sigma) - Phi((v - 1/2) / sigma)] / [2 Phi(19.5 / sigma) - 1],
so P(|e| <= i) = erf((i + 1/2) / (sigma sqrt 2)) / erf(19.5 / (sigma sqrt 2)).  Entry i (0 <= i <= 18) is
floor(2^63 P(|e| <= i)); the sampler takes the top 63 bits x of a 64-bit draw, |e| = #{i : x >= cdt[i]}, sign = low bit.
usage: python tools/noise_cdt.py        prints the 19 constants (tests/test_encrypt_sampler.py compares both copies with them)"""
from decimal import Decimal, getcontext

getcontext().prec = 90
SIGMA = Decimal("3.19")
BOUND = 19


def erf(x):
    # Maclaurin series 2/sqrt(pi) sum (-1)^n x^(2n+1) / (n! (2n+1)); |x| < 4.4 here: terms peak near 1e8, 90 digits leave > 60
    term, total, n = x, x, 0
    while abs(term) > Decimal(10) ** -85:
        n += 1
        term = -term * x * x / n
        total += term / (2 * n + 1)
    pi = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798")
    return 2 * total / pi.sqrt()


def table():
    s2 = SIGMA * Decimal(2).sqrt()
    norm = erf((Decimal(BOUND) + Decimal("0.5")) / s2)
    return [int((erf((Decimal(i) + Decimal("0.5")) / s2) / norm * (1 << 63)).to_integral_value(rounding="ROUND_FLOOR")) for i in range(BOUND)]


if __name__ == "__main__":
    for i, v in enumerate(table()):
        print("0x%016xULL,  /* P(|e| <= %d) */" % (v, i))
