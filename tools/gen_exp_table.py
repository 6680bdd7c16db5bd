#!/usr/bin/env python
"""Constants of codegen.py's table-driven float64 exp (exp_tbl64): 2^(j/64) correctly rounded,
ln2/64 split for an exact k*C1, and the degree-3 near-minimax polynomial of
(e^r - 1 - r) / r^2 on |r| <= ln2/128 (interpolation at Chebyshev nodes, 60-digit arithmetic).
Prints C++ hex-float literals; codegen.py carries the output verbatim."""
from decimal import Decimal, getcontext
from fractions import Fraction
import math

getcontext().prec = 70
LN2 = Decimal(2).ln()


def dexp(x):
    return Decimal(x).exp()


def to_double(d):           # correctly rounded (float() of a decimal string rounds correctly)
    return float(str(d))


def main():
    tbl = [to_double(Decimal(2) ** (Decimal(j) / 64)) for j in range(64)]
    inv = to_double(Decimal(64) / LN2)
    c = LN2 / 64
    # C1: 33 significant bits of ln2/64 (|k| < 2^17, so k*C1 is exact in double)
    f = Fraction(to_double(c))
    e = math.frexp(float(f))[1]
    scale = Fraction(2) ** (33 - e)
    c1 = float(Fraction(round(Fraction(str(c)) * scale), 1) / scale)
    c2 = to_double(c - Decimal(c1))
    a = LN2 / 128
    n = 4
    pi = Decimal("3.14159265358979323846264338327950288419716939937510582097494")

    def dcos(x):            # Taylor, 60 digits
        s, t, k = Decimal(1), Decimal(1), 0
        while abs(t) > Decimal(10) ** -65:
            k += 2
            t = -t * x * x / (k * (k - 1))
            s += t
        return s
    nodes = [a * dcos(pi * (2 * i + 1) / (2 * n)) for i in range(n)]

    def g(r):
        return (dexp(r) - 1 - r) / (r * r)
    # solve the Vandermonde system exactly in Decimal (Gaussian elimination)
    A = [[r ** k for k in range(n)] + [g(r)] for r in nodes]
    for i in range(n):
        p = max(range(i, n), key=lambda r_: abs(A[r_][i]))
        A[i], A[p] = A[p], A[i]
        for r_ in range(i + 1, n):
            m = A[r_][i] / A[i][i]
            A[r_] = [x - m * y for x, y in zip(A[r_], A[i])]
    co = [Decimal(0)] * n
    for i in reversed(range(n)):
        co[i] = (A[i][n] - sum(A[i][k] * co[k] for k in range(i + 1, n))) / A[i][i]
    cs = [to_double(x) for x in co]
    # max error of the polynomial over the interval (relative to e^r)
    worst = Decimal(0)
    for i in range(-2000, 2001):
        r = a * i / 2000
        if r == 0:
            continue
        p = r + r * r * sum(Decimal(cs[k]) * r ** k for k in range(n))
        worst = max(worst, abs((1 + p) - dexp(r)) / dexp(r))
    print("// polynomial error (relative, exact arithmetic): %.3e = %.4f ulp" % (worst, worst / Decimal(2) ** -53))
    print("INV  = %s  // 64/ln2" % inv.hex())
    print("C1   = %s  // ln2/64, 33 bits" % c1.hex())
    print("C2   = %s" % c2.hex())
    for k, v in enumerate(cs):
        print("P%d   = %s  // ~1/%d!" % (k + 2, v.hex(), k + 2))
    print("static const double tbl[64] = {")
    for j in range(0, 64, 4):
        print("  " + ", ".join(v.hex() for v in tbl[j:j + 4]) + ",")
    print("};")


if __name__ == "__main__":
    main()
