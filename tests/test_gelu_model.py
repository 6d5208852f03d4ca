"""CPU: the GELU constants of gigapose_amd/csrc/gp_common.h (gp_gelu_scaled: erfc(|x| / sqrt 2) = 2^-Q(min(|x|, 9)), one fma for both signs) read
from the header and evaluated in emulated f32 arithmetic (tools/fit_gelu.py) against float64 -- pins the constants the HIP epilogues compile
(the GPU side of the same bound: tests/test_gpu_split.py::test_gelu_epilogue_on_a_grid_vs_float64)."""
import os
import re
import sys

import numpy as np
from scipy.special import ndtr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fit_gelu  # noqa: E402


def header_coefficients():
    src = open(os.path.join(ROOT, "gigapose_amd", "csrc", "gp_common.h")).read()
    body = src[src.index("float gp_gelu_scaled(float x, float hs)"):]
    body = body[:body.index("return")]
    first = float(re.search(r"float p = ([-0-9.e+]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"fmaf\(p, z, ([-0-9.e+]+)f\)", body)]
    return list(reversed([first] + rest))   # c1 .. c8


def test_header_constants_give_the_documented_error():
    c = header_coefficients()
    assert len(c) == 8
    x = np.concatenate([np.linspace(-12, 12, 240001), np.array([0.0, -0.0, 20.0, -20.0, 1e3, -1e3, 1e-20])])
    want = x.astype(np.float32).astype(np.float64)
    want = want * ndtr(want)
    for hs in (0.5, 4.0):    # GELU itself; the x 8 activation planes' scale folded in (a power of two: the same bits)
        got = fit_gelu.gelu_new(x, c, hs).astype(np.float64) / (2.0 * hs)
        err = np.abs(got - want) / (np.abs(x) + 1.0)
        assert err.max() < 5.5e-8, err.max()
    assert np.array_equal(fit_gelu.gelu_new(x, c, 0.5) * np.float32(8.0), fit_gelu.gelu_new(x, c, 4.0))
    # the packed two-element form of the header carries the same constants
    src = open(os.path.join(ROOT, "gigapose_amd", "csrc", "gp_common.h")).read()
    two = src[src.index("gp_gelu_scaled2("):]
    two = two[:two.index("return r;")]
    consts = [float(v) for v in re.findall(r"k\(([-0-9.e+]+)f\)", two)]
    assert list(reversed(consts)) == c


def test_fit_reproduces_the_header_constants():
    m, c = fit_gelu.fit(8, n=60001, iters=60)      # a coarser run of the committed fit
    assert m < 1e-8
    x = np.linspace(-12, 12, 120001)
    a, b = fit_gelu.gelu_new(x, c), fit_gelu.gelu_new(x, header_coefficients())
    assert (np.abs(a.astype(np.float64) - b.astype(np.float64)) / (np.abs(x) + 1.0)).max() < 1e-7   # an ulp of the result (2.4e-7 at |x| in [2, 4))
