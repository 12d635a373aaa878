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


def report_params(name, got, want, steps, lr, rtol=1e-4, atol=3e-5, loose_frac=5e-4):
    """Parameters after `steps` Adam updates.  Adam's update is lr * m / (sqrt(v) + eps): for an element whose
    gradient is near zero it behaves like lr * sign(g), so a gradient difference at rounding level can move that
    element by up to 2 * lr in one step (on ANY two hosts, the reference's included).  Hence: every element within
    2 * lr * steps, and all but a `loose_frac` fraction within the tight bound."""
    got = np.asarray(got.detach().cpu().double()); want = np.asarray(want.detach().cpu().double())
    assert got.shape == want.shape
    err = np.abs(got - want)
    bad = ~(err <= atol + rtol * np.abs(want))
    msg = f"{name}: max|err|={np.nanmax(err):.3e} outside tight bound {int(bad.sum())}/{bad.size}"
    print(msg)
    assert not np.isnan(err).any(), msg
    assert err.max() <= 2.0 * lr * steps + atol, msg
    assert bad.sum() <= max(1, int(loose_frac * bad.size)), msg


def report_after_updates(name, got, want, rtol, atol, steps, lr, loose_frac=1e-4):
    """An activation computed from parameters that went through `steps` Adam updates: the few weights report_params
    lets move by O(lr) (gradient ~ eps) shift the activations they feed by O(lr * |input|_1).  Step 0 is held to the
    tight bound everywhere; later steps keep the tight bound on all but a `loose_frac` fraction, and every element
    within 20 * lr * steps relative to the largest activation."""
    if steps == 0:
        return report(name, got, want, rtol, atol)
    got = np.asarray(got.detach().cpu().double()); want = np.asarray(want.detach().cpu().double())
    assert got.shape == want.shape
    err = np.abs(got - want)
    bad = ~(err <= atol + rtol * np.abs(want))
    msg = f"{name}: max|err|={np.nanmax(err):.3e} outside tight bound {int(bad.sum())}/{bad.size}"
    print(msg)
    assert not np.isnan(err).any(), msg
    assert err.max() <= atol + 20.0 * lr * steps * np.abs(want).max(), msg
    assert bad.sum() <= max(1, int(loose_frac * bad.size)), msg


def cuda_batch(b):
    out = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
    if "noise" in b:
        out["noise"] = [t.to(DEV) for t in b["noise"]]
    if "dropmask" in b:
        dm = b["dropmask"]
        out["dropmask"] = None if dm[0] is None else torch.stack(dm).to(DEV).contiguous()
    return out


class ModuleRegions:
    """debug_region() of a cmlpl_amd.models.BaseNet2 forward: the workspace the autograd node of its output keeps
    (one network), addressed like TrainEngine.debug_region."""
    debug_nets = 1

    def __init__(self, net, out, n):
        import ctypes as C
        from cmlpl_amd import _lib
        self._C, self._lib, self.net, self.n = C, _lib, net, n
        self.ws = out.grad_fn.saved_tensors[5]
        assert self.ws.dtype == torch.uint8

    def debug_region(self, name, dtype=torch.float32):
        C = self._C
        off, nbytes = C.c_size_t(), C.c_size_t()
        self._lib.check("cmlpl_debug_region", self._lib.load().cmlpl_debug_region(
            C.byref(self.net._cshape), 1, self.n, name.encode(), C.byref(off), C.byref(nbytes)))
        return self.ws[off.value: off.value + nbytes.value].view(dtype)


def hip_relu_gates(eng, shape, n):
    """The ReLU decisions the HIP forward took, as the oracle's ``relu_gates`` (per network {"z1","z2","zy"} bool
    tensors in the oracle's NCHW layout), read from the saved masks m1 / m2 (u8 per pooled element, bit
    (h&1)*2+(w&1)) and the spectral ReLU output y.  Pixels the floor-pooling drops (last row / column of an odd
    map) are never computed on the device and never reach a gradient: their gate is left True."""
    H2, W2 = shape.H // 2, shape.W // 2
    H4, W4 = H2 // 2, W2 // 2
    out = []
    nets = getattr(eng, "debug_nets", 2)
    for net in range(nets):
        g = {}
        for name, key, hh, ww, HH, WW in (("m1", "z1", H2, W2, shape.H, shape.W), ("m2", "z2", H4, W4, H2, W2)):
            m = eng.debug_region(name, torch.uint8).view(nets, n, hh, ww, 64)[net].cpu()
            gate = torch.ones(n, 64, HH, WW, dtype=torch.bool)
            for dh in range(2):
                for dw in range(2):
                    bit = ((m >> (dh * 2 + dw)) & 1).bool()                    # [n, hh, ww, 64]
                    gate[:, :, dh:2 * hh:2, dw:2 * ww:2] = bit.permute(0, 3, 1, 2)
            g[key] = gate
        g["zy"] = eng.debug_region("y").view(nets, n, 1024)[net].cpu() > 0
        out.append(g)
    return out


def relu_mask_audit(eng, taps, shape, n, ztol=2e-5, ztol_y=None):
    """Compare the ReLU masks the HIP forward saved (workspace regions m1, m2, y) with the signs of
    the oracle's pre-activations.  fp32 summation order differs between the two, so an element whose
    pre-activation is ~1 ulp from zero may legitimately land on the other side; every mismatch must
    therefore sit at |z_oracle| < ztol -- anything else is a kernel bug.  Returns flips[net][layer].
    `ztol_y` (default ztol) bounds the spectral branch separately: after Adam updates a single feat_spe weight whose
    gradient is ~eps may sit O(lr) away from the oracle's (see report_params), which shifts that output column by
    O(lr * |x|)."""
    ztol_y = ztol if ztol_y is None else ztol_y
    H2, W2 = shape.H // 2, shape.W // 2
    H4, W4 = H2 // 2, W2 // 2
    out = []
    nets = getattr(eng, "debug_nets", 2)
    for net in range(nets):
        res = {}
        for name, zkey, hh, ww in (("m1", "z1", H2, W2), ("m2", "z2", H4, W4)):
            m = eng.debug_region(name, torch.uint8).view(nets, n, hh * ww, 64)[net].cpu()
            z = taps[net][zkey]                                  # [n, 64, Hfull, Wfull]
            cnt, worst = 0, 0.0
            for dh in range(2):
                for dw in range(2):
                    zz = z[:, :, dh:2 * hh:2, dw:2 * ww:2].permute(0, 2, 3, 1).reshape(n, hh * ww, 64)
                    bit = ((m >> (dh * 2 + dw)) & 1).bool()
                    diff = bit != (zz > 0)
                    if diff.any():
                        cnt += int(diff.sum())
                        worst = max(worst, float(zz[diff].abs().max()))
            assert worst < ztol, f"net {net} {name}: ReLU mask differs at |z|={worst:.3e} (not a rounding flip)"
            res[zkey] = cnt
        y = eng.debug_region("y").view(nets, n, 1024)[net].cpu()
        zy = taps[net]["zy"]
        diff = (y > 0) != (zy > 0)
        worst = float(zy[diff].abs().max()) if diff.any() else 0.0
        assert worst < ztol_y, f"net {net} y: ReLU mask differs at |z|={worst:.3e}"
        res["zy"] = int(diff.sum())
        out.append(res)
    return out


def sync_engine_from_oracle(eng, st):
    """Put the engine on the oracle's trajectory again (used after a legitimate ReLU-boundary flip)."""
    for net in range(2):
        for k in O.LIVE_KEYS:
            eng.view(eng.params, net, k).copy_(st.params[net][k])
            eng.view(eng.m, net, k).copy_(st.adam[net].m[k])
            eng.view(eng.v, net, k).copy_(st.adam[net].v[k])
    eng._packed_dirty = True
