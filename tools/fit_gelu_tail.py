"""Coefficients of the GELU kernel arithmetic in memvul_amd/csrc/common.h.

gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|), and log2 Phi(-a) is smooth enough on a >= 0 that a degree-7 polynomial
Q(a) (Chebyshev least squares on [0, 6.5]) gives Phi(-a) = exp2(Q(a)) to 9e-6 relative, i.e. the exact-erf GELU of HF
BertIntermediate to 6e-7 absolute with ONE transcendental per element; the leading coefficient is negative and Q
decreases monotonically beyond the fit interval, so no clamp is needed (|x| exp2(Q(|x|)) < 3e-10 for |x| > 6.5).

    python tools/fit_gelu_tail.py      # prints the float32 coefficients and the error report
"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

A, DEG = 6.5, 7


def fit():
    n = 6000
    xs = np.cos(np.pi * (np.arange(n) + 0.5) / n) * A / 2 + A / 2
    c = C.chebfit((xs - A / 2) / (A / 2), np.log2(0.5 * erfc(xs / np.sqrt(2))), DEG)
    pc = C.cheb2poly(c)
    poly, t = np.poly1d([0.0]), np.poly1d([2.0 / A, -1.0])
    for k, ck in enumerate(pc):
        poly = poly + ck * (t ** k)
    return [float(np.float32(v)) for v in poly.c[::-1]]  # ascending powers of a


def gelu_fp32(x, co):
    """The kernel's operation sequence in float32 (Horner with fma-equivalent rounding is within 1 ulp of this)."""
    x = x.astype(np.float32)
    az = np.abs(x)
    q = np.full_like(az, np.float32(co[-1]))
    for c in co[-2::-1]:
        q = q * az + np.float32(c)
    e = np.exp2(q.astype(np.float64)).astype(np.float32)
    s = x * np.float32(0.5) + az * np.float32(0.5)
    return s - az * e


if __name__ == "__main__":
    co = fit()
    print("coefficients (a^0 .. a^7):", co)
    x = np.linspace(-12, 12, 2400001)
    ref = x * 0.5 * erfc(-x / np.sqrt(2))
    print("max |gelu - exact| on [-12, 12]:", float(np.abs(gelu_fp32(x, co) - ref).max()))
