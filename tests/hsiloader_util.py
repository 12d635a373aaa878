"""Tiny on-disk dataset in the layout sample_generation.py writes (XP.npy, X.npy, Y.npy, train/unlabel/test index
arrays), built from a seed: shared by the fixture generator (which runs the REFERENCE HSIDataSet on it) and the
CPU test (which runs this repo's drop-in on the same files)."""
import os

import numpy as np

CASES = (   # (name, setindex, max_iters, num_unlabel)
    ("label_tiled", "label", 23, 1000),       # 10 labelled samples tiled to 23: 2 repeats + 3 (hsi_loader.py:28-33)
    ("label_plain", "label", None, 1000),
    ("unlabel_cut_tiled", "unlabel", 30, 7),  # first 7 of the unlabelled indices, tiled to 30 (:36-45)
    ("unlabel_all", "unlabel", None, 1000),   # num_unlabel beyond the array: slicing clamps
    ("unlabel_exact", "unlabel", 25, 25),     # max_iters == len: one repeat, empty remainder
    ("test", "test", None, 1000),
    ("wholeset", "wholeset", None, 1000),
)


def make_tiny_dataset(root, seed=7, n=40, C=3, w=4, bands=5, K=4):
    os.makedirs(root, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(seed))
    np.save(os.path.join(root, "XP.npy"), rng.standard_normal((n, C, w, w)))           # float64 on disk, like the
    np.save(os.path.join(root, "X.npy"), rng.standard_normal((n, bands)))              # reference's preprocessing
    np.save(os.path.join(root, "Y.npy"), rng.integers(1, K + 1, size=n).astype(np.uint8))   # labels are 1-based
    perm = rng.permutation(n)
    np.save(os.path.join(root, "train_array.npy"), perm[:10])
    np.save(os.path.join(root, "unlabel_array.npy"), perm[3:28])      # overlaps the labelled set, as in the reference
    np.save(os.path.join(root, "test_array.npy"), perm[28:])
