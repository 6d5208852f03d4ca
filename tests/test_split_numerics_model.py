"""CPU model of the `split` numerics: f32 operands as two f16 planes, three exact f16 x f16
products per k, f32 accumulation.  numpy restates what the HIP kernels compute (gp_split256.hip: single accumulator,
operands pre-scaled by powers of two; gp_vit.hip attention_split_kernel: the same for S = Q K^T and P V with P split as
2^15 p) and pins the ALGORITHM's error against float64 -- the GPU tests then pin the kernels against the same bounds.
No GPU, no oracle import: this is arithmetic only."""
import numpy as np


def split(x, scale):
    """hi = f16(scale x), lo = f16(scale x - hi); returned as float64 (f16 x f16 products are exact in f32 / f64)."""
    v = (np.asarray(x, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def split_matmul(a, b, sa, sb):
    """sum_k a[i,k] b[j,k] the way gemm_planes256_kernel forms it: hi hi + hi lo + lo hi, one accumulator, exact rescale."""
    ah, al = split(a, sa)
    bh, bl = split(b, sb)
    acc = ah @ bh.T + ah @ bl.T + al @ bh.T  # float64 here; the kernel accumulates in f32 (tested on the GPU)
    return acc / (sa * sb)


def test_plane_representation_keeps_22_bits():
    rng = np.random.RandomState(0)
    x = rng.randn(100000) * np.exp(rng.uniform(-3, 3, 100000))
    hi, lo = split(x, 8.0)
    rel = np.abs((hi + lo) / 8.0 - x.astype(np.float32)) / np.abs(x)
    big = np.abs(8.0 * x) >= 2.0 ** -3   # lo is a normal f16 there
    assert rel[big].max() < 2.0 ** -21
    # below, lo is quantised to f16's subnormal spacing 2^-24 (MI355X's MFMA honours f16 subnormals, tools/probe_attn_split.py)
    assert np.abs((hi + lo) / 8.0 - x.astype(np.float32))[~big].max() <= 2.0 ** -25 / 8.0 * 1.0001


def test_split_gemm_error_is_f32_class():
    rng = np.random.RandomState(1)
    K = 1024
    w = rng.randn(64, K) * 0.03     # weight-like
    x = rng.randn(96, K) * 1.5      # activation-like
    got = split_matmul(w, x, 64.0, 8.0)
    ref = w.astype(np.float32).astype(np.float64) @ x.astype(np.float32).astype(np.float64).T
    mag = np.abs(w) @ np.abs(x).T
    err = np.abs(got - ref) / mag
    assert err.max() < 2e-7          # dropped lo lo term + 2^-22 representation; the f32 fmaf chain measures 3.9e-7
    # a plain f16 matmul (hi planes only) is 3 orders worse: why the low halves are carried
    hi_only = (split(w, 64.0)[0] @ split(x, 8.0)[0].T) / 512.0
    assert (np.abs(hi_only - ref) / mag).max() > 1e-5


def attention_split_model(q, k, v):
    """attention_split_kernel's arithmetic for one head (gp_vit.hip): planes x 8, S rescaled by 2^-9, P as 2^15 p = hi + lo."""
    qh, ql = split(q, 8.0)
    kh, kl = split(k, 8.0)
    vh, vl = split(v, 8.0)
    s = (qh @ kh.T + qh @ kl.T + ql @ kh.T) * (0.125 / 64.0)
    p = np.exp(s - s.max(axis=1, keepdims=True)).astype(np.float32)
    ph, pl = split(p, 32768.0)
    acc = ph @ vh + pl @ vh + ph @ vl
    return acc / p.astype(np.float64).sum(axis=1, keepdims=True) / 32768.0 / 8.0


def test_split_attention_model_matches_float64():
    rng = np.random.RandomState(2)
    worst = 0.0
    for scale in (0.3, 1.5, 3.0):   # diffuse ... sharply peaked softmax
        q, k, v = rng.randn(257, 64) * scale, rng.randn(257, 64) * scale, rng.randn(257, 64)
        deq = [sum(split(t, 8.0)) / 8.0 for t in (q, k, v)]
        s = deq[0] @ deq[1].T * 0.125
        p = np.exp(s - s.max(axis=1, keepdims=True))
        ref = (p / p.sum(axis=1, keepdims=True)) @ deq[2]
        got = attention_split_model(q, k, v)
        worst = max(worst, np.abs(got - ref).max() / np.abs(ref).max())
    assert worst < 5e-7, worst       # the kernel measures 7.1e-7 (f32 accumulation + 2-ulp exp on top)


def test_fused_multiply_convert_pitfall():
    """Why plane producers pin the rounded product: hi from ONE rounding of the exact product and lo
    from the f32-ROUNDED product disagree by an f16 ulp of hi at ties -- an error of the size of lo itself."""
    rng = np.random.RandomState(3)
    a = rng.rand(400000).astype(np.float32) + np.float32(1.0)
    b = np.float32(0.7853982)
    exact = a.astype(np.float64) * np.float64(b)
    v32 = (a * b).astype(np.float32)                        # what lo is computed from
    hi_fused = exact.astype(np.float16)                     # v_fma_mixlo_f16: single rounding of the exact product
    hi_two_step = v32.astype(np.float16)                    # (f16)(f32 product)
    lo = (v32 - hi_fused.astype(np.float32)).astype(np.float16)
    consistent = hi_fused.astype(np.float64) + lo.astype(np.float64)
    assert np.abs(consistent - v32).max() < 2.0 ** -20      # a CONSISTENT pair is fine whichever rounding made hi ...
    mixed = hi_two_step.astype(np.float64) + lo.astype(np.float64)   # ... storing the other hi with that lo is not
    bad = np.abs(mixed - v32) > 2.0 ** -12
    assert bad.any() and bad.mean() < 1e-3                  # rare (ties only), but an f16 ulp of hi when it happens


def test_a_lower_plane_scale_only_raises_the_subnormal_floor():
    """The error model behind the per-tensor plane scales (round 5): with scale s, an element keeps 22 bits while
    |s x| >= 2^-3 (lo a normal f16); below that lo is quantised to f16's subnormal spacing, an ABSOLUTE error of 2^-25 / s.  Going from
    s = 8 to s = 1 (a tensor holding an outlier of ~1e4) therefore leaves a GEMM over activation-like values f32-class: the dot product's
    error grows by the floor term only, far below the f32 chain's 3.9e-7 of sum |a||b|."""
    rng = np.random.RandomState(2)
    x = rng.randn(200000) * np.exp(rng.uniform(-4, 2, 200000))
    for s in (8.0, 1.0, 0.125):
        hi, lo = split(x, s)
        err = np.abs((hi + lo) / s - x.astype(np.float32))
        big = np.abs(s * x) >= 2.0 ** -3
        assert (err[big] / np.abs(x[big])).max() < 2.0 ** -21
        assert err[~big].max() <= 2.0 ** -25 / s * 1.0001
    K = 4096
    w = rng.randn(32, K) * 0.03
    act = np.maximum(rng.randn(48, K), 0.0) * 0.8      # GELU-like: half zeros, a right tail
    act[:, 7] = 1.2e4                                   # the massive unit that forces s = 1
    ref = w.astype(np.float32).astype(np.float64) @ act.astype(np.float32).astype(np.float64).T
    mag = np.abs(w) @ np.abs(act).T
    e1 = (np.abs(split_matmul(w, act, 64.0, 1.0) - ref) / mag).max()
    assert e1 < 2e-7, e1
    # with the default s = 8 the same tensor does not fit f16 at all: the outlier's hi plane is inf
    with np.errstate(over="ignore"):
        assert not np.isfinite(split(act, 8.0)[0]).all()
