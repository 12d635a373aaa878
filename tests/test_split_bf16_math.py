"""CPU: the arithmetic claim behind "fp32 on the bf16 MFMA" (cmlpl_amd/csrc/common.hpp b3_split),
restated in numpy: an fp32 value is EXACTLY the sum of three bf16 pieces obtained by successive truncation, and the six
piece-products of weight >= 2^-16 reproduce the exact product to 2^-21 relative in the worst case (truncated pieces
all carry the sign of the value, so each residual can approach a full unit of its piece's last place) and about
2^-24 -- one fp32 rounding -- on average."""
import numpy as np


def trunc16(x):
    """top 16 bits of an fp32 value (a bf16 number held in fp32)"""
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(v):
    p1 = trunc16(v)
    r1 = (v - p1).astype(np.float32)          # exact in fp32
    p2 = trunc16(r1)
    r2 = (r1 - p2).astype(np.float32)
    p3 = trunc16(r2)
    return p1, p2, p3, r1, r2


def _values(n, seed):
    g = np.random.default_rng(seed)
    v = (g.standard_normal(n) * np.exp(g.uniform(-20, 20, n))).astype(np.float32)
    v[:16] = np.float32([0, -0.0, 1, -1, 1.5, 3.1415927, 1e-30, -1e30, 2 ** -126, 65504, 0.1, -0.3, 7e-12, 1 + 2 ** -23,
                         2 - 2 ** -22, 123456.789])
    return v


def test_three_truncated_pieces_are_exact():
    v = _values(200000, 1)
    p1, p2, p3, r1, r2 = split3(v)
    # residuals are exact: no rounding in v - p1 and r1 - p2
    assert np.array_equal(r1.astype(np.float64), v.astype(np.float64) - p1.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - p2.astype(np.float64))
    # every piece is a bf16 number (low 16 bits clear), the third one captures what is left
    for p in (p1, p2, p3):
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))
    assert np.array_equal(p3, r2)
    s = p1.astype(np.float64) + p2.astype(np.float64) + p3.astype(np.float64)
    assert np.array_equal(s, v.astype(np.float64))


def test_six_products_match_the_exact_product_to_a_few_ulps():
    a, b = _values(100000, 2), _values(100000, 3)
    keep = (np.abs(a) < 1e15) & (np.abs(b) < 1e15) & (np.abs(a) > 1e-15) & (np.abs(b) > 1e-15)   # no overflow / underflow
    a, b = a[keep], b[keep]
    a1, a2, a3, _, _ = split3(a)
    b1, b2, b3, _, _ = split3(b)
    f = lambda x: x.astype(np.float64)
    six = f(a1) * f(b1) + (f(a1) * f(b2) + f(a2) * f(b1)) + (f(a1) * f(b3) + f(a2) * f(b2) + f(a3) * f(b1))
    exact = f(a) * f(b)
    rel = np.abs(six - exact) / np.abs(exact)
    # what is dropped: a2 b3 + a3 b2 + a3 b3, with |x2| < 2^-7 |x| and |x3| < 2^-15 |x|
    assert rel.max() < 2.0 ** -21, rel.max()
    assert rel.mean() < 2.0 ** -23, rel.mean()
    # each bf16 x bf16 product has a 16-bit significand: exact in fp32, as the MFMA forms it
    prod = (a1 * b1).astype(np.float32)
    assert np.array_equal(f(prod), f(a1) * f(b1))
