"""Offline fit of the GELU the split kernels' epilogues evaluate (gigapose_amd/csrc/gp_common.h: gp_gelu_scaled), with its error model.

    erfc(|x| / sqrt 2) = 2^-Q(z),  z = min(|x|, 9),  Q(z) = z (c1 + c2 z + ... + c8 z^7)
    2 hs GELU(x) = (a + |a|) - |a| 2^-Q,  a = hs x                  (one fma for both signs)

Q is a weighted minimax fit (Lawson iterations on a least-squares problem) of -log2 erfc; the weight makes the error of GELU uniform in
|x| + 1.  The script prints the coefficients and, in emulated f32 arithmetic (fma = one rounding, v_exp_f32 modelled as correctly rounded),
max |error| / (|x| + 1) and the rms error on N(0, 2) inputs for this form, for the round-2 form it replaced (erfc = t P9(t) exp(-z^2),
t = 1 / (1 + 0.3275911 z)) and for f32 0.5 x (1 + erf) with an exact erf.  CPU only (numpy, scipy)."""
import numpy as np
from scipy.special import erf, log_ndtr, ndtr

f32, LN2, XM = np.float32, np.log(2.0), 9.0


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.asarray(c, dtype=np.float64)).astype(f32)


def fit(deg, n=300001, iters=200):
    xs = np.linspace(0.0, XM, n)
    q = -(1.0 + log_ndtr(-xs) / LN2)                       # -log2 erfc(x / sqrt 2) = -log2(2 Phi(-x))
    w = xs * 0.5 * 2.0 ** -q * LN2 / (xs + 1.0) + 1e-30    # d GELU / d Q over (|x| + 1)
    A = np.stack([xs ** k for k in range(1, deg + 1)], 1)
    ww, best = w.copy(), None
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(A * ww[:, None], q * ww, rcond=None)
        err = np.abs((A @ c - q) * w)
        if best is None or err.max() < best[0]:
            best = (err.max(), c.copy())
        ww = ww * (1.0 + err / err.max())
    return best


def gelu_new(x, c, hs=0.5):
    x = x.astype(f32)
    z = np.minimum(np.abs(x), f32(XM))
    p = np.full_like(z, f32(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        p = fma(p, z, f32(c[k]))
    e = np.exp2(-(p * z).astype(f32).astype(np.float64)).astype(f32)
    a = (f32(hs) * x).astype(f32)
    return fma(-np.abs(a), e, (a + np.abs(a)).astype(f32))


def gelu_round2(x):
    x = x.astype(f32)
    z = (np.abs(x) * f32(0.70710678118654752440)).astype(f32)
    t = (f32(1) / fma(np.full_like(z, f32(0.3275911)), z, f32(1.0))).astype(f32)
    cs = [0.02651038324816011, -0.284242067130941, 0.8274675429718052, -0.8329329722108324, 0.7896692586773671, -0.14005398441300523,
          0.25548806601570556, 0.17245176856740801, 0.18564199446374482]
    p = np.full_like(z, f32(cs[0]))
    for k in cs[1:]:
        p = fma(p, t, f32(k))
    c = ((p * t).astype(f32) * np.exp((-(z * z).astype(f32)).astype(np.float64)).astype(f32)).astype(f32)
    return ((f32(0.5) * x).astype(f32) * np.where(x >= 0, (f32(2) - c).astype(f32), c)).astype(f32)


if __name__ == "__main__":
    xt = np.concatenate([np.linspace(-12, 12, 1200001), np.random.RandomState(0).randn(1000000) * 2])
    ex = xt.astype(f32).astype(np.float64)
    ex = ex * ndtr(ex)

    def report(name, g):
        e = np.abs(g.astype(np.float64) - ex)
        r = e / (np.abs(xt) + 1)
        print(f"{name}: max |err| / (|x| + 1) {r.max():.3e} at x = {xt[r.argmax()]:.3f}; rms error on N(0, 2) inputs {np.sqrt((e[-1000000:] ** 2).mean()):.3e}")

    report("round-2 form, t P9(t) exp(-z^2)      ", gelu_round2(xt))
    report("f32 0.5 x (1 + erf), erf exact       ", (f32(0.5) * xt.astype(f32) * (1 + erf((xt.astype(f32) * f32(0.70710678)).astype(np.float64)).astype(f32))).astype(f32))
    for deg in (6, 7, 8, 9):
        m, c = fit(deg)
        report(f"2^-Q form, degree {deg} (fit error {m:.1e})", gelu_new(xt, c))
        if deg == 8:
            print("   c1..c8 =", ", ".join(f"{v:.17g}" for v in c))
