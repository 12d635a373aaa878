#!/usr/bin/env python3
"""Golden vectors for patch extraction: run the reference's own
``ExtractPatches`` (tools/hyper_tools.py:226-243) on seeded cubes and store inputs-by-seed + outputs.
Build container only:  python tests/golden/make_golden_patches.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden._refload import load_reference_module  # noqa: E402

ht = load_reference_module("tools/hyper_tools.py", "ref_hyper_tools")
out = {}
for name, (row, col, C, w, seed) in {"a": (9, 7, 5, 4, 51), "b": (6, 11, 3, 6, 52), "c": (23, 22, 4, 20, 53)}.items():
    X = np.random.Generator(np.random.PCG64(seed)).standard_normal((row, col, C)).astype(np.float32)
    ref = ht.ExtractPatches(X, w)                      # [K, C, w, w]
    idx = np.array([0, col - 1, (row - 1) * col, row * col - 1, (row // 2) * col + col // 2, 1, col])
    out[name + "_cfg"] = np.array([row, col, C, w, seed])
    out[name + "_idx"] = idx
    out[name + "_patches"] = ref[idx]
    out[name + "_sum"] = np.array([ref.astype(np.float64).sum(), np.abs(ref).astype(np.float64).sum()])
np.savez_compressed(os.path.join(HERE, "patches_ref.npz"), **out)
print({k: v.shape for k, v in out.items() if k.endswith("_patches")})
