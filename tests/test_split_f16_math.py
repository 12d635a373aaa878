"""CPU: the arithmetic claims behind "fp32 as TWO fp16 pieces" (cmlpl_amd/csrc/common.hpp h_split / h2_split_w / mfma_h2,
DESIGN.md section 4), restated in numpy: x s = h1 + h2 + e with h1 = fp16(x s) truncated, h2 = fp16 of the EXACT residual
truncated; the packing side rounds to nearest; a . w ~ h1 g1 + h2 g1 + h1 g2; the power-of-two scales and their inverse are
exact for every exponent the kernels accept."""
import numpy as np


def rtz16(x):
    """fp32 -> fp16 rounded toward zero (v_cvt_pkrtz_f16_f32), as float64; |x| < 65520"""
    h = x.astype(np.float16)
    up = np.abs(h.astype(np.float64)) > np.abs(x.astype(np.float64))
    h = np.where(up, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float64)


def rne16(x):
    return x.astype(np.float16).astype(np.float64)


def a_pieces(x, sc):
    """h_split: scaled value (exact: a power of two), truncated first piece, exact residual, truncated second piece"""
    xs = (x * sc).astype(np.float32)
    h1 = rtz16(xs)
    r = (xs - h1.astype(np.float32)).astype(np.float32)
    assert np.array_equal(r.astype(np.float64), xs.astype(np.float64) - h1)          # the residual is exact in fp32
    h2 = rtz16(r)
    return xs.astype(np.float64), h1, h2


def w_pieces(w):
    """h2_split_w: w 2^13, nearest first piece, nearest piece of the exact residual"""
    ws = (w * np.float32(8192.0)).astype(np.float32)
    g1 = rne16(ws)
    r = (ws - g1.astype(np.float32)).astype(np.float32)
    assert np.array_equal(r.astype(np.float64), ws.astype(np.float64) - g1)
    g2 = rne16(r)
    return ws.astype(np.float64), g1, g2


def _sample_scale(x):
    """the kernels' per-sample scale: exponent field e of max |x| -> 2^(14 - (e - 127)), inverse incl. the weights' 2^13"""
    e = int(np.float32(np.abs(x).max()).view(np.uint32) >> 23)
    assert 40 <= e <= 200
    sc = np.uint32((268 - e) << 23).view(np.float32)
    inv = np.uint32((e - 27) << 23).view(np.float32)
    return sc, inv


def test_scales_are_exact_powers_of_two_over_the_whole_accepted_range():
    for e in range(40, 201):
        m = np.uint32((e << 23) | 0x7FFFFF).view(np.float32)          # the largest value with that exponent
        sc = np.uint32((268 - e) << 23).view(np.float32)
        inv = np.uint32((e - 27) << 23).view(np.float32)
        assert 2.0 ** 14 <= float(m) * float(sc) < 2.0 ** 15 < 65504.0
        assert float(sc) * float(inv) * 8192.0 == 1.0                  # 1 / (sc 2^13), exactly
        assert float(np.float32(m) * sc) == float(m) * float(sc)       # scaling does not round


def test_two_pieces_carry_the_value_to_2_pow_minus_21_and_the_three_products_to_2_pow_minus_20():
    g = np.random.default_rng(5)
    n = 400000
    # activations of one "sample": magnitudes spread over 2^17 below the maximum (where both pieces are normal fp16 numbers)
    x = (g.standard_normal(n) * np.exp2(g.uniform(-16, 0, n))).astype(np.float32)
    x[:6] = np.float32([1.9999999, -1.9999999, 1 + 2 ** -23, 2 ** -16, 3.0 * 2 ** -17, -(2 - 2 ** -11)])   # mantissas of ones: worst truncations
    w = (g.standard_normal(n) * 0.05 * np.exp2(g.uniform(-6, 3, n))).astype(np.float32)
    w = np.clip(w, -7.9, 7.9)
    sc, inv = _sample_scale(x)
    xs, h1, h2 = a_pieces(x, sc)
    ws, g1, g2 = w_pieces(w)
    keep = (np.abs(xs) >= 2.0 ** -3) & (np.abs(ws) >= 2.0 ** -3)        # residuals normal: the scheme's full-precision range
    ea = np.abs(xs - h1 - h2)[keep] / np.abs(xs)[keep]
    ew = np.abs(ws - g1 - g2)[keep] / np.abs(ws)[keep]
    assert ea.max() < 2.0 ** -21 and ew.max() <= 2.0 ** -22, (ea.max(), ew.max())
    prod = h1 * g1 + h2 * g1 + h1 * g2                                  # the three MFMAs' exact products
    exact = xs * ws
    rel = (np.abs(prod - exact) / np.abs(exact))[keep]
    print(f"two-piece product: worst {rel.max():.3e} (2^{np.log2(rel.max()):.2f}), mean {rel.mean():.3e} (2^{np.log2(rel.mean()):.2f})")
    assert rel.max() < 2.0 ** -20, rel.max()                            # truncations 2^-21 + 2^-22, the dropped h2 g2 < 2^-21
    assert rel.mean() < 2.0 ** -22.5, rel.mean()
    # each fp16 x fp16 product has a 22-bit significand: exact in the MFMA's fp32 accumulate input
    p32 = (h1.astype(np.float32) * g1.astype(np.float32)).astype(np.float64)
    assert np.array_equal(p32, h1 * g1)
    # un-scaling is exact
    assert np.array_equal((exact * float(inv)), x.astype(np.float64) * w.astype(np.float64))


def test_elements_far_below_the_sample_maximum_degrade_gracefully():
    """an element 2^-k of the maximum keeps 22 - max(0, k - 17) bits: absolute error bounded by 2^-24 in scaled units"""
    x = np.float32([1.0] + [1.2345678 * 2.0 ** -k for k in range(1, 36)])
    sc, _ = _sample_scale(x)
    xs, h1, h2 = a_pieces(x, sc)
    err = np.abs(xs - h1 - h2)
    assert (err <= 2.0 ** -24).all() or (err / np.abs(xs) < 2.0 ** -21).all()
    for k in range(1, 36):
        rel = err[k] / abs(xs[k])
        assert rel < 2.0 ** -(21 - max(0, k - 17)) * 1.0001, (k, rel)


def test_weight_range_flag_condition():
    """h2_split_w's flag: |w| 2^13 > 65000, or not finite"""
    w = np.float32([7.9, -7.93, 7.94, 8.0, np.inf, np.nan, 0.0, 1e-30])
    bad = ~(np.abs(w * np.float32(8192.0)) <= 65000.0)
    assert bad.tolist() == [False, False, True, True, True, True, False, False]
