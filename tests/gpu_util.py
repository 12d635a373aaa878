"""Shared helpers for the GPU parity tests (HIP path vs the CPU oracle)."""
import numpy as np
import torch

from oracle import cmlpl_oracle as O
from cmlpl_amd import NetShape, HyperParams

DEV = "cuda:0"


def to_shape(s: "O.NetShape") -> NetShape:
    return NetShape(s.C, s.H, s.W, s.bands, s.K)


def to_hp(h: "O.HyperParams") -> HyperParams:
    return HyperParams(**{k: getattr(h, k) for k in HyperParams.__dataclass_fields__})


def report(name, got, want, rtol, atol):
    got = np.asarray(got.detach().cpu().double() if torch.is_tensor(got) else got, dtype=np.float64)
    want = np.asarray(want.detach().cpu().double() if torch.is_tensor(want) else want, dtype=np.float64)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    if got.size == 0:
        return 0.0
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = ~(err <= tol)           # catches NaN as well
    worst = np.unravel_index(np.argmax(np.where(np.isnan(err), np.inf, err - tol)), err.shape)
    msg = (f"{name}: max|err|={np.nanmax(err):.3e} max|want|={np.max(np.abs(want)):.3e} "
           f"bad={int(bad.sum())}/{bad.size} worst@{worst} got={got[worst]:.6e} want={want[worst]:.6e}")
    print(msg)
    assert not bad.any(), msg
    return float(np.nanmax(err))


def cuda_batch(b):
    out = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
    if "noise" in b:
        out["noise"] = [t.to(DEV) for t in b["noise"]]
    if "dropmask" in b:
        dm = b["dropmask"]
        out["dropmask"] = None if dm[0] is None else torch.stack(dm).to(DEV).contiguous()
    return out
