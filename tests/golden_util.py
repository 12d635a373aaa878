"""Helpers shared by the golden-fixture tests (CPU oracle and GPU parity)."""
import glob
import os

import numpy as np
import torch

from oracle import cmlpl_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    """step-trajectory fixtures (patches_ref.npz belongs to the patch-extraction row, N3)"""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("patches", "ntxent", "losshelper"))]


class GoldenCase:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        c = self.z["cfg"]
        self.shape = O.NetShape(int(c[0]), int(c[1]), int(c[2]), int(c[3]), int(c[4]))
        self.bt, self.btu, self.steps, self.seed = int(c[5]), int(c[6]), int(c[7]), int(c[8])
        self.epoch0, self.batch_index0, self.num_epochs = int(c[9]), int(c[10]), int(c[11])
        f = self.z["cfg_f"]
        self.dropout, self.cls_scale, self.separable, self.thr = map(float, f)
        self.full_steps = [int(s) for s in self.z["full_steps"]]
        self.hp = O.HyperParams(num_epochs=self.num_epochs, thr=self.thr, dropout=self.dropout)

    def params(self):
        p0 = O.closed_form_params(self.shape, self.seed)
        p1 = O.closed_form_params(self.shape, self.seed + 1)
        if self.cls_scale != 1.0:
            p0["classifier.weight"] *= self.cls_scale
            p1["classifier.weight"] *= self.cls_scale
        return p0, p1

    def batch(self, s):
        return O.synthetic_batch(self.shape, self.bt, self.btu, self.seed * 1000 + s,
                                 dropout=self.dropout, separable=self.separable)

    def epoch_bi(self, s):
        return self.epoch0, self.batch_index0 + s


def rel_close(a, b, rtol, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def rel_err(a, b, floor=1e-12):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor))) if a.size else 0.0
