"""CPU: the counter-based generators behind the in-kernel augmentation noise (cmlpl_amd/csrc/common.hpp: pcg4d +
noise_normal4 for the spectra -- four normals per hash call, 32-bit uniforms -- and noise_normal8 for the patches --
EIGHT normals per call: each 32-bit hash word gives a 16-bit radius uniform and a 16-bit angle), restated in numpy with the same integer arithmetic and the same Box-Muller form (exact log / sin /
cos here where the kernel uses the hardware's v_log_f32 / v_sin_f32 / v_cos_f32 -- a last-bit difference per value,
nothing a distribution test sees).  Checks what the reference's torch.randn draws have (train.py:157-182): zero mean,
unit variance, Gaussian kurtosis and tails, no correlation between neighbouring elements, between the four normals of
one hash call, between the two networks' streams, between consecutive steps and between samples."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return (x & M32).astype(np.uint64)


def pcg4d(x, y, z, w):
    x = _u32(x * np.uint64(1664525) + np.uint64(1013904223)); y = _u32(y * np.uint64(1664525) + np.uint64(1013904223))
    z = _u32(z * np.uint64(1664525) + np.uint64(1013904223)); w = _u32(w * np.uint64(1664525) + np.uint64(1013904223))
    x = _u32(x + y * w); y = _u32(y + z * x); z = _u32(z + x * y); w = _u32(w + y * z)
    x ^= x >> np.uint64(16); y ^= y >> np.uint64(16); z ^= z >> np.uint64(16); w ^= w >> np.uint64(16)
    x = _u32(x + y * w); y = _u32(y + z * x); z = _u32(z + x * y); w = _u32(w + y * z)
    return x, y, z, w


def noise_normal4(seed, step, stream, ctr):
    """ctr: uint64 array (noise_ctr: (global sample << 24) | group); returns [4, len(ctr)] float32 normals"""
    seed, step = np.uint64(seed), np.uint64(step)
    s0, s1 = seed & M32, seed >> np.uint64(32)
    ctr = ctr.astype(np.uint64)
    x = (ctr & M32) ^ s0
    y = (ctr >> np.uint64(32)) ^ s1
    z = np.full_like(ctr, _u32(np.uint64(stream) * np.uint64(0x9E3779B9)) ^ (step >> np.uint64(32)) ^ _u32(s1 * np.uint64(0x85EBCA6B)))
    w = np.full_like(ctr, (step & M32) ^ _u32(s0 * np.uint64(0xC2B2AE35)))
    rx, ry, rz, rw = pcg4d(x, y, z, w)
    k = np.float32(2.3283064365386963e-10)
    f = lambda a: a.astype(np.float32)          # (float)uint32, round to nearest like v_cvt_f32_u32
    u0 = np.minimum(f(rx) * k + k, np.float32(1.0)); u2 = np.minimum(f(rz) * k + k, np.float32(1.0))
    r0 = np.sqrt(np.float32(-1.3862943611198906) * np.log2(u0)); r1 = np.sqrt(np.float32(-1.3862943611198906) * np.log2(u2))
    a0 = f(ry) * k; a1 = f(rw) * k                # angle in revolutions
    tp = np.float32(2 * np.pi)
    return np.stack([r0 * np.cos(tp * a0), r0 * np.sin(tp * a0), r1 * np.cos(tp * a1), r1 * np.sin(tp * a1)]).astype(np.float32)


def noise_hash_words(seed, step, stream, ctr):
    seed, step = np.uint64(seed), np.uint64(step)
    s0, s1 = seed & M32, seed >> np.uint64(32)
    ctr = ctr.astype(np.uint64)
    x = (ctr & M32) ^ s0
    y = (ctr >> np.uint64(32)) ^ s1
    z = np.full_like(ctr, _u32(np.uint64(stream) * np.uint64(0x9E3779B9)) ^ (step >> np.uint64(32)) ^ _u32(s1 * np.uint64(0x85EBCA6B)))
    w = np.full_like(ctr, (step & M32) ^ _u32(s0 * np.uint64(0xC2B2AE35)))
    return pcg4d(x, y, z, w)


def noise_normal8(seed, step, stream, ctr):
    """ctr: uint64 array (noise_ctr: (global sample << 24) | PAIR index c); returns [8, len(ctr)] float32 normals of elements
    8c .. 8c + 7: hash word i -> elements 2i (radius * cos) and 2i + 1 (radius * sin), radius from the word's high 16 bits
    (u = (hi + 1) / 2^16), angle = low 16 bits / 2^16 revolutions"""
    k16 = np.float32(1.52587890625e-05)
    tp = np.float32(2 * np.pi)
    out = []
    for wd in noise_hash_words(seed, step, stream, ctr):
        hi = (wd >> np.uint64(16)).astype(np.float32); lo = (wd & np.uint64(0xFFFF)).astype(np.float32)
        u = hi * k16 + k16
        r = np.sqrt(np.float32(-1.3862943611198906) * np.log2(u))
        a = lo * k16
        out += [r * np.cos(tp * a), r * np.sin(tp * a)]
    return np.stack(out).astype(np.float32)


def _ctr(sample, groups):
    return (np.uint64(sample) << np.uint64(24)) | groups.astype(np.uint64)


def test_moments_tails_and_independence():
    groups = np.arange(3116)                    # one PaviaU patch: 103 * 121 = 12463 elements = 3116 groups of four
    z = np.concatenate([noise_normal4(1088, 7, 0x100, _ctr(s, groups)).T.reshape(-1) for s in range(96)])   # ~1.2 M
    n = z.size
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs((z ** 4).mean() - 3.0) < 5 * np.sqrt(96.0 / n)
    for t, p in ((1.0, 0.31731), (2.0, 0.045500), (3.0, 0.0026998)):      # two-sided Gaussian tail mass
        got = (np.abs(z) > t).mean()
        assert abs(got - p) < 5 * np.sqrt(p * (1 - p) / n), (t, got, p)
    assert np.abs(z).max() < 6.7                                           # u >= 2^-32 -> |z| <= sqrt(64 ln 2) = 6.66
    # neighbours in memory (the four normals of a call, and across calls)
    for lag in (1, 2, 3, 4, 121):
        c = np.corrcoef(z[:-lag], z[lag:])[0, 1]
        assert abs(c) < 5 / np.sqrt(n), (lag, c)


def test_patch_generator_moments_tails_and_independence():
    """noise_normal8 (the patches): same checks; the tail stops at sqrt(2 ln 2^16) = 4.71 (mass beyond: 2.5e-6)"""
    pairs = np.arange(1558)                     # one PaviaU patch: 12463 elements = 1558 pairs of 16-byte groups
    z = np.concatenate([noise_normal8(1088, 7, 0x100, _ctr(s, pairs)).T.reshape(-1) for s in range(96)])   # ~1.2 M
    n = z.size
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs((z ** 4).mean() - 3.0) < 5 * np.sqrt(96.0 / n)
    for t, p in ((1.0, 0.31731), (2.0, 0.045500), (3.0, 0.0026998), (4.0, 6.334e-5)):
        got = (np.abs(z) > t).mean()
        assert abs(got - p) < 5 * np.sqrt(p * (1 - p) / n), (t, got, p)
    assert np.abs(z).max() <= np.sqrt(32 * np.log(2.0)) + 1e-5             # u >= 2^-16 -> |z| <= 4.7096
    for lag in (1, 2, 3, 4, 7, 8, 121):         # inside a word's pair, across words, across calls, across a band row
        c = np.corrcoef(z[:-lag], z[lag:])[0, 1]
        assert abs(c) < 5 / np.sqrt(n), (lag, c)
    # radius and angle of a word are independent: z_even^2 + z_odd^2 (the radius) against atan2 (the angle)
    r2 = z[0::2] ** 2 + z[1::2] ** 2
    ang = np.arctan2(z[1::2], z[0::2])
    assert abs(np.corrcoef(r2, ang)[0, 1]) < 5 / np.sqrt(r2.size)
    base = noise_normal8(1088, 7, 0x100, _ctr(5, pairs)).reshape(-1)
    for name, o in {"other network": noise_normal8(1088, 7, 0x101, _ctr(5, pairs)).reshape(-1),
                    "next step": noise_normal8(1088, 8, 0x100, _ctr(5, pairs)).reshape(-1),
                    "next sample": noise_normal8(1088, 7, 0x100, _ctr(6, pairs)).reshape(-1),
                    "other seed": noise_normal8(1089, 7, 0x100, _ctr(5, pairs)).reshape(-1)}.items():
        assert abs(np.corrcoef(base, o)[0, 1]) < 5 / np.sqrt(base.size), name


def test_streams_steps_and_samples_are_independent():
    groups = np.arange(3116)
    base = noise_normal4(1088, 7, 0x100, _ctr(5, groups)).reshape(-1)
    n = base.size
    others = {
        "other network (stream + 1)": noise_normal4(1088, 7, 0x101, _ctr(5, groups)).reshape(-1),
        "spectral stream": noise_normal4(1088, 7, 0x200, _ctr(5, groups)).reshape(-1),
        "next step": noise_normal4(1088, 8, 0x100, _ctr(5, groups)).reshape(-1),
        "next sample": noise_normal4(1088, 7, 0x100, _ctr(6, groups)).reshape(-1),
        "other seed": noise_normal4(1089, 7, 0x100, _ctr(5, groups)).reshape(-1),
    }
    for name, o in others.items():
        assert not np.array_equal(o, base), name
        c = np.corrcoef(base, o)[0, 1]
        assert abs(c) < 5 / np.sqrt(n), (name, c)
    # counter-based: the same (seed, step, stream, sample, group) gives the same four normals wherever it is formed
    again = noise_normal4(1088, 7, 0x100, _ctr(5, groups[100:200]))
    assert np.array_equal(again, noise_normal4(1088, 7, 0x100, _ctr(5, groups))[:, 100:200])
