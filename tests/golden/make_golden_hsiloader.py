#!/usr/bin/env python3
"""Fixture for the loader surface: run the REFERENCE hsi_loader.HSIDataSet (hsi_loader.py:5-133,
loaded by file path) on a tiny seeded dataset and store what it returns -- length and every item tuple -- for the
constructor variants train.py uses (train.py:101-114).  Build container only:
    python tests/golden/make_golden_hsiloader.py
Only the outputs are stored; the test rebuilds the same tiny dataset from the seed."""
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden._refload import load_reference_module  # noqa: E402
from tests.hsiloader_util import CASES, make_tiny_dataset  # noqa: E402


def main():
    ref = load_reference_module("hsi_loader.py", "ref_hsi_loader")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        make_tiny_dataset(os.path.join(tmp, "dataset", "PaviaU"))
        cwd = os.getcwd()
        os.chdir(tmp)                       # the reference opens './dataset/PaviaU/' relative to the cwd
        try:
            for name, setindex, max_iters, num_unlabel in CASES:
                ds = ref.HSIDataSet(1, setindex=setindex, max_iters=max_iters, num_unlabel=num_unlabel)
                items = [ds[i] for i in range(len(ds))]
                out[name + "_len"] = np.asarray(len(ds))
                out[name + "_XP"] = np.stack([it[0] for it in items])
                out[name + "_X"] = np.stack([it[1] for it in items])
                assert out[name + "_XP"].dtype == np.float32 and out[name + "_X"].dtype == np.float32
                if setindex != "wholeset":
                    assert all(len(it) == 3 for it in items)
                    out[name + "_Y"] = np.asarray([int(it[2]) for it in items], dtype=np.int64)
                else:
                    assert all(len(it) == 2 for it in items)
                print(name, len(ds), out[name + "_XP"].shape)
        finally:
            os.chdir(cwd)
    path = os.path.join(HERE, "hsiloader_ref.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


if __name__ == "__main__":
    main()
