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
    return [n for n in names if not n.startswith(("patches", "ntxent", "losshelper", "hsiloader"))]


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
        self.dead = tuple(int(v) for v in self.z["cfg_dead"]) if "cfg_dead" in self.z.files else (-1, -1)

    def params(self):
        p0 = O.closed_form_params(self.shape, self.seed)
        p1 = O.closed_form_params(self.shape, self.seed + 1)
        if self.cls_scale != 1.0:
            p0["classifier.weight"] *= self.cls_scale
            p1["classifier.weight"] *= self.cls_scale
        if self.dead != (-1, -1):
            for p in (p0, p1):
                p["feat_spe.bias"] = -p["feat_spe.bias"].abs()
        return p0, p1

    def batch(self, s):
        b = O.synthetic_batch(self.shape, self.bt, self.btu, self.seed * 1000 + s,
                              dropout=self.dropout, separable=self.separable)
        if self.dead != (-1, -1):
            O.kill_spectral_rows([], b, self.dead[0], self.dead[1])
        return b

    def epoch_bi(self, s):
        return self.epoch0, self.batch_index0 + s


def rel_close(a, b, rtol, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def rel_err(a, b, floor=1e-12):
    """max relative error; NaN/inf must sit at the same places with the same sign (then they count as equal),
    otherwise the result is inf."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    fin = np.isfinite(a) & np.isfinite(b)
    same_nonfinite = (np.isnan(a) & np.isnan(b)) | ((a == b) & ~fin)
    if not np.all(fin | same_nonfinite):
        return float("inf")
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(a[fin] - b[fin]) / (np.abs(b[fin]) + floor)))
