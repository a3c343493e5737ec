"""The GELU of the FFN-1 epilogue (memvul_amd/csrc/common.h: gelu_erf / gelu_erf2) is max(x,0) - |x| exp2(Q(|x|)) with a
degree-7 polynomial Q fitted to log2 Phi(-a) (tools/fit_gelu_tail.py).  CPU check of that arithmetic, restated in
float32 numpy with the coefficients read from the header, against the exact erf GELU of HF BertIntermediate."""
import os
import re

import numpy as np
from scipy.special import erfc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_coefficients():
    src = open(os.path.join(ROOT, "memvul_amd", "csrc", "common.h")).read()
    co = [float(re.search(r"#define MV_GELU_Q%d \(([-0-9.e+]+)f\)" % k, src).group(1)) for k in range(8)]
    return co


def test_header_coefficients_are_the_committed_fit():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fit_gelu_tail
    assert np.allclose(header_coefficients(), fit_gelu_tail.fit(), rtol=2e-6, atol=0)


def test_gelu_polynomial_form_matches_exact_erf_gelu():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fit_gelu_tail
    co = header_coefficients()
    x = np.concatenate([np.linspace(-40, 40, 800001), [0.0, -0.0, 1e-30, -1e-30, 6.5, -6.5, 100.0, -100.0, 3e4, -3e4]])
    x = x.astype(np.float32).astype(np.float64)  # the kernel sees fp32 inputs
    got = fit_gelu_tail.gelu_fp32(x, co).astype(np.float64)
    ref = x * 0.5 * erfc(-x / np.sqrt(2))
    assert np.isfinite(got).all()
    assert (np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max() < 1e-6
    # where the value is stored as fp16 the approximation is far inside the rounding step
    big = np.abs(ref) > 1e-4
    assert (np.abs(got - ref)[big] / np.abs(ref)[big]).max() < 5e-5
    # Q has no bump beyond the fit interval: the tail term only shrinks
    a = np.linspace(6.5, 300, 30000).astype(np.float64)
    q = sum(c * a ** k for k, c in enumerate(co))
    assert (np.diff(q) < 0).all() and (a * np.exp2(q)).max() < 1e-9
