"""Patch extraction.  CPU: the oracle restatement vs vectors produced by the reference's
own ExtractPatches (tests/golden/make_golden_patches.py).  GPU: the HIP gather vs the oracle, bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import cmlpl_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patches_ref.npz")


def _cube(row, col, C, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((row, col, C)).astype(np.float32)


def test_oracle_matches_reference_extract_patches():
    z = np.load(GOLD)
    for name in ("a", "b", "c"):
        row, col, C, w, seed = (int(v) for v in z[name + "_cfg"])
        X = _cube(row, col, C, seed)
        full = O.extract_patches(X, w)
        assert full.shape == (row * col, C, w, w)
        assert np.array_equal(full[z[name + "_idx"]], z[name + "_patches"])            # exact
        assert np.allclose([full.astype(np.float64).sum(), np.abs(full).astype(np.float64).sum()], z[name + "_sum"],
                           rtol=1e-12)
        idx = z[name + "_idx"]
        assert np.array_equal(O.extract_patches(X, w, idx), z[name + "_patches"])      # subset path


def test_mirror_is_symmetric_edge_repeating():
    assert O.mirror_index(np.array([-3, -1, 0, 4, 5, 7]), 5).tolist() == [2, 0, 0, 4, 4, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("row,col,C,w", [(9, 7, 5, 4), (23, 22, 4, 20), (31, 17, 103, 11), (40, 33, 60, 20),
                                         (12, 50, 200, 11), (25, 25, 48, 15)])
def test_hip_extract_patches_bit_exact(row, col, C, w):
    from cmlpl_amd.patches import extract_patches
    X = _cube(row, col, C, 100 + w)
    rng = np.random.Generator(np.random.PCG64(7))
    idx = np.concatenate([np.array([0, col - 1, (row - 1) * col, row * col - 1]),       # the four corners
                          rng.integers(0, row * col, size=61)])
    want = O.extract_patches(X, w, idx)
    got = extract_patches(torch.from_numpy(X).cuda(), torch.from_numpy(idx).cuda(), w).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
