"""The arithmetic of the f16x3 GEMM (re-net_amd/csrc/gemm_h3.h) restated in numpy -- TEST INFRASTRUCTURE, CPU only:
tensor scale from a magnitude bound, two binary16 planes per operand, three exact f16 x f16 products accumulated in
fp32, cross terms scaled by 2^-11 once.  Checks the error claims the header makes, independently of any GPU: every
element is represented to 2^-22 relative (two 11-bit significands) at any tensor magnitude, GEMM results stay in the
class of plain fp32 products, a bound 2^10 too large only costs binades, elements below 2^-29 max |x| keep an absolute
error of 2^-39 max |x|.  (The GPU tests compare the kernel itself with fp64 and with
the exact-fp32 MFMA kernel: tests/test_gpu_parity.py.)"""
import numpy as np
import pytest


def scale_of(bound):
    """2^(15 - e) for bound = m 2^e, m in [0.5, 1)  (h3_scale_of)."""
    if not np.isfinite(bound) or bound <= 0:
        return 2.0 ** 126
    _, e = np.frexp(np.float32(bound))
    return float(2.0 ** min(max(15 - int(e), -126), 126))


def split(x, s):
    xs = (x.astype(np.float32) * np.float32(s)).astype(np.float32)
    h1 = xs.astype(np.float16)
    h2 = ((xs - h1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return h1, h2


def f16x3_matmul(a, b, bound_a=None, bound_b=None):
    """a [M, K] @ b [N, K]^T as the kernel computes it (fp32 accumulation by numpy's float32 matmul)."""
    sa = scale_of(np.abs(a).max() if bound_a is None else bound_a)
    sb = scale_of(np.abs(b).max() if bound_b is None else bound_b)
    a1, a2 = split(a, sa)
    b1, b2 = split(b, sb)
    f = lambda h: h.astype(np.float32)                                        # noqa: E731
    main = f(a1) @ f(b1).T
    corr = f(a1) @ f(b2).T + f(a2) @ f(b1).T
    return ((main + corr * np.float32(2.0 ** -11)).astype(np.float64) / sa) / sb


def units(a, b, c):
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    unit = (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T) * 2.0 ** -24
    return np.abs(c - ref) / unit


@pytest.mark.parametrize('sa,sb', [(1.0, 1.0), (3e-9, 7e4), (5e7, 2e-6), (1e-20, 1e-12), (1e15, 1e10)])
def test_split_products_are_fp32_class_at_any_tensor_magnitude(sa, sb):
    rng = np.random.RandomState(9)
    m, n, k = 96, 80, 700
    a = rng.standard_normal((m, k)) * np.exp2(rng.uniform(-20, 0, (m, 1)))
    b = rng.standard_normal((n, k)) * np.exp2(rng.uniform(-20, 0, (1, k)))
    a, b = (a * sa).astype(np.float32), (b * sb).astype(np.float32)
    err = units(a, b, f16x3_matmul(a, b)).max()
    f32 = units(a, b, (a @ b.T).astype(np.float64)).max()                     # plain fp32 products and accumulation
    # numpy's float32 matmul has its own accumulation order: the yardstick is the same product in plain fp32
    assert err <= 12.0 and err <= 2.0 * max(f32, 0.5), (err, f32)


def test_split_is_exact_to_2_pow_minus_24_and_a_loose_bound_costs_only_binades():
    rng = np.random.RandomState(1)
    x = (rng.standard_normal(20000) * np.exp2(rng.uniform(-27, 0, 20000))).astype(np.float32)
    for slack in (1.0, 1024.0):
        s = scale_of(np.abs(x).max() * slack)
        h1, h2 = split(x, s)
        back = (h1.astype(np.float64) + h2.astype(np.float64) / 2048.0) / s
        big = np.abs(x) >= np.abs(x).max() * slack * 2.0 ** -29
        assert big.sum() > 1000
        rel = np.abs(back[big] - x[big].astype(np.float64)) / np.abs(x[big])
        assert rel.max() <= 2.0 ** -22 * 1.5, (slack, rel.max())              # u^2 with u = 2^-11, + the subnormal edge of h2
        assert np.abs(back - x.astype(np.float64)).max() <= 2.0 ** -22 * np.abs(x).max() * slack
        assert np.isfinite(h1.astype(np.float32)).all() and np.isfinite(h2.astype(np.float32)).all()
    # far below the bound: binary16 subnormals, absolute error bounded by the scaled subnormal spacing
    tiny = (rng.standard_normal(2000) * 2.0 ** -34).astype(np.float32)
    s = scale_of(1.0)
    h1, h2 = split(tiny, s)
    back = (h1.astype(np.float64) + h2.astype(np.float64) / 2048.0) / s
    assert np.abs(back - tiny.astype(np.float64)).max() <= 2.0 ** -39


def test_scale_brings_the_bound_into_the_top_binade_and_never_overflows():
    for bound in (1e-30, 3e-7, 0.5, 1.0, 1.5, 65504.0, 1e20):
        s = scale_of(bound)
        assert 2.0 ** 14 <= bound * s < 2.0 ** 15, (bound, s)
        x = np.array([bound, -bound], np.float32)
        h1, h2 = split(x, s)
        assert np.isfinite(h1.astype(np.float32)).all() and np.isfinite(h2.astype(np.float32)).all()
    assert scale_of(0.0) == 2.0 ** 126
