"""CPU oracle for the CMLPL training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU (fp32) restatement of the reference's per-step
training computation.  It is *the checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``cmlpl_amd``) never imports anything from ``oracle/`` and
fails loudly when the HIP library is missing.

Pinning: ``tests/golden/make_golden.py`` runs the *reference itself*
(``/root/reference/tools/models.py:BaseNet2`` imported, ``train.py:150-279``
text-exec'd on CPU) on closed-form seeded inputs and stores its outputs as
small ``.npz`` fixtures; ``tests/test_oracle_golden.py`` checks this restatement
against those fixtures.  Parity is therefore pinned by outputs of the reference
run in the build container (the reference ships no tests or golden vectors of
its own for this path -- SURVEY.md section 4).

Reference lines restated (all under /root/reference/):
  * BaseNet2.forward ............ tools/models.py:130-152 (Normalize :87-90)
  * input augmentation .......... train.py:157-158,163-164,170-171,181-182
  * supervised CE / accuracy .... train.py:191-194,278
  * pseudo-label smoothing ...... train.py:203-219
  * adaptive threshold .......... train.py:147-148,220-222,227-228
  * memory bank update .......... train.py:138-145,223-237 (incl. the ptr1 quirk :237)
  * mutual soft-target CE ....... train.py:239-242
  * similarity softmax .......... train.py:243-247,257-258
  * pseudo-label graph .......... train.py:249-256
  * contrastive loss ............ train.py:260-265
  * totals / backward / Adam .... train.py:131-132,266-272
  * logged row .................. train.py:274-278
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

FEAT_DIM = 1024          # tools/models.py:119 (n_fc1)
CONV_CH = 64             # tools/models.py:102-107


@dataclass(frozen=True)
class NetShape:
    """Shape parameters of one BaseNet2.

    The reference hard-codes C=60 and a 2624-wide classifier
    (tools/models.py:102,127), which only fits 20..23-pixel windows; the
    generalisation derives both from (C, H, W).
    """
    C: int = 60          # input channels of conv0 (PCA components / bands)
    H: int = 20          # window height
    W: int = 20          # window width
    bands: int = 103     # spectral vector length (num_features)
    K: int = 9           # classes

    @property
    def H2(self): return self.H // 2          # AvgPool2d(2,2) floors (models.py:110)
    @property
    def W2(self): return self.W // 2
    @property
    def H4(self): return self.H2 // 2
    @property
    def W4(self): return self.W2 // 2
    @property
    def spatial_feat(self): return CONV_CH * self.H4 * self.W4
    @property
    def cls_in(self): return self.spatial_feat + FEAT_DIM


@dataclass
class HyperParams:
    """Flags of train.py:356-379 plus the literals hard-coded in the step."""
    lr: float = 5e-4             # train.py:365
    num_epochs: int = 20         # train.py:366
    thr: float = 1.0             # train.py:369
    alpha: float = 0.95          # train.py:371
    queue_batch: float = 17      # train.py:372
    temperature: float = 0.3     # train.py:374
    dropout: float = 0.8         # train.py:377
    noise: float = 0.5           # train.py:378
    w_contrast: float = 0.5      # literal, train.py:266,270
    w_mutual: float = 4.0        # literal, train.py:266,270
    pos_thr: float = 0.8         # literal, train.py:251
    neg_thr: float = 0.3         # literal, train.py:254
    bank_step: int = 256         # literal pointer advance, train.py:234,237
    bank_mult: int = 5           # queue_size = 5 * labeled_batch_size * 2, train.py:138
    beta1: float = 0.9           # torch.optim.Adam defaults (train.py:131)
    beta2: float = 0.999
    eps: float = 1e-8

    def adap_thr(self, epoch: int) -> float:
        # train.py:147-148  (numpy float64 scalar)
        decay = epoch / self.num_epochs
        return float(np.exp(-0.5 * (decay ** 2)))


# --------------------------------------------------------------------------- #
# parameters
# --------------------------------------------------------------------------- #
LIVE_KEYS = ("conv0.weight", "conv0.bias", "conv1.weight", "conv1.bias",
             "conv2.weight", "conv2.bias", "feat_spe.weight", "feat_spe.bias",
             "classifier.weight", "classifier.bias")


def param_shapes(s: NetShape) -> "OrderedDict[str, tuple]":
    """The 16 state_dict keys of BaseNet2 in registration order (models.py:98-128)."""
    return OrderedDict([
        ("conv0.weight", (CONV_CH, s.C, 1, 1)), ("conv0.bias", (CONV_CH,)),
        ("conv1.weight", (CONV_CH, CONV_CH, 3, 3)), ("conv1.bias", (CONV_CH,)),
        ("conv2.weight", (CONV_CH, CONV_CH, 3, 3)), ("conv2.bias", (CONV_CH,)),
        ("feat_spe.weight", (FEAT_DIM, s.bands)), ("feat_spe.bias", (FEAT_DIM,)),
        ("feat_ss.weight", (256, FEAT_DIM)), ("feat_ss.bias", (256,)),
        ("feat_ss2.weight", (64, FEAT_DIM)), ("feat_ss2.bias", (64,)),
        ("feat_ss3.weight", (64, 256)), ("feat_ss3.bias", (64,)),
        ("classifier.weight", (s.K, s.cls_in)), ("classifier.bias", (s.K,)),
    ])


def closed_form_params(s: NetShape, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic parameters from numpy PCG64 (no torch RNG, so fixtures do not
    depend on the torch version).  Same distribution as torch's default init:
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    shapes = param_shapes(s)
    for key, shp in shapes.items():
        mod = key.split(".")[0]
        wshape = shapes[mod + ".weight"]
        fan_in = int(np.prod(wshape[1:]))
        b = 1.0 / math.sqrt(fan_in)
        out[key] = torch.from_numpy(rng.uniform(-b, b, size=shp).astype(np.float32))
    return out


# --------------------------------------------------------------------------- #
# BaseNet2.forward  (tools/models.py:130-152)
# --------------------------------------------------------------------------- #
def l2norm(x: torch.Tensor) -> torch.Tensor:
    # Normalize.forward, models.py:87-90 -- no epsilon (0/0 -> NaN is reference behaviour)
    norm = x.pow(2).sum(1, keepdim=True).pow(0.5)
    return x.div(norm)


def _relu(z: torch.Tensor, gate: Optional[torch.Tensor]) -> torch.Tensor:
    """ReLU; with ``gate`` (bool, same shape) the activation pattern is GIVEN instead of derived from sign(z):
    out = z where gate else 0, d out / d z = gate.  Tests use it to make the oracle take the device's ReLU
    decisions where a pre-activation sits within rounding of zero (fp32 summation order decides the sign there),
    so that gradients can be compared at a tight tolerance; every gate/sign disagreement is separately required
    to sit at |z| < 2e-5 (tests/gpu_util.py)."""
    if gate is None:
        return F.relu(z)
    return torch.where(gate, z, torch.zeros_like(z))


def basenet2_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, y: torch.Tensor,
                     dropmask: Optional[torch.Tensor] = None, taps: Optional[dict] = None,
                     relu_gates: Optional[dict] = None):
    """x: [n,C,H,W], y: [n,bands].  ``dropmask`` is the explicit dropout
    multiplier ([n, cls_in], values 0 or 1/(1-p)); None = eval / p==0.
    ``relu_gates``: optional {"z1","z2","zy"} -> bool activation patterns (see _relu).
    Returns (logits [n,K], feat [n,1024])."""
    g = relu_gates or {}
    x = F.conv2d(x, p["conv0.weight"], p["conv0.bias"])                    # :132
    x_res = x
    x = F.conv2d(x, p["conv1.weight"], p["conv1.bias"], padding=1)         # :134
    if taps is not None: taps["z1"] = (x + x_res).detach()                 # pre-ReLU, for mask audits
    x = _relu(x + x_res, g.get("z1"))                                       # :135
    x = F.avg_pool2d(x, 2, 2)                                               # :136
    x_res = x
    x = F.conv2d(x, p["conv2.weight"], p["conv2.bias"], padding=1)         # :138
    if taps is not None: taps["z2"] = (x + x_res).detach()
    x = _relu(x + x_res, g.get("z2"))                                       # :139
    x = F.avg_pool2d(x, 2, 2)                                               # :140
    x = x.reshape(x.size(0), -1)                                            # :141
    y = F.linear(y, p["feat_spe.weight"], p["feat_spe.bias"])              # :142
    if taps is not None: taps["zy"] = y.detach()
    y = _relu(y, g.get("zy"))                                               # :143
    cat = torch.cat([x, y], 1)                                              # :144
    feat = l2norm(y)                                                        # :145-146
    if dropmask is not None:
        cat = cat * dropmask                                                # :147-148
    logits = F.linear(cat, p["classifier.weight"], p["classifier.bias"])   # :150
    return logits, feat


# --------------------------------------------------------------------------- #
# training state
# --------------------------------------------------------------------------- #
@dataclass
class AdamState:
    m: Dict[str, torch.Tensor]
    v: Dict[str, torch.Tensor]
    t: int = 0


@dataclass
class StepState:
    shape: NetShape
    params: List[Dict[str, torch.Tensor]]            # [Base, Base1]
    adam: List[AdamState]
    bank_feats: List[torch.Tensor]                   # [queue_feats, queue_feats1]
    bank_probs: List[torch.Tensor]                   # [queue_probs, queue_probs1]
    ptr: List[int] = field(default_factory=lambda: [0, 0])

    @staticmethod
    def create(shape: NetShape, params0, params1, labeled_batch_size: int,
               hp: Optional[HyperParams] = None) -> "StepState":
        hp = hp or HyperParams()
        q = hp.bank_mult * labeled_batch_size * 2                           # train.py:138,142
        ps = [OrderedDict((k, v.clone()) for k, v in params0.items()),
              OrderedDict((k, v.clone()) for k, v in params1.items())]
        adam = [AdamState({k: torch.zeros_like(v) for k, v in pp.items() if k in LIVE_KEYS},
                          {k: torch.zeros_like(v) for k, v in pp.items() if k in LIVE_KEYS})
                for pp in ps]
        return StepState(shape, ps, adam,
                         [torch.zeros(q, FEAT_DIM), torch.zeros(q, FEAT_DIM)],   # :139,143
                         [torch.zeros(q, shape.K), torch.zeros(q, shape.K)])     # :140,144


def adam_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
                t: int, hp: HyperParams) -> None:
    """torch.optim.Adam single-tensor update (defaults: no amsgrad, no weight
    decay), in place; ``t`` is the 1-based step count."""
    m.lerp_(g, 1.0 - hp.beta1)
    v.mul_(hp.beta2).addcmul_(g, g, value=1.0 - hp.beta2)
    bc1 = 1.0 - hp.beta1 ** t
    bc2 = 1.0 - hp.beta2 ** t
    step_size = hp.lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(hp.eps)
    p.addcdiv_(m, denom, value=-step_size)


def bank_write(bank: torch.Tensor, ptr: int, rows: torch.Tensor) -> None:
    """queue[ptr:ptr+n] = rows (train.py:232-236), written modulo the bank size
    where the reference's slice-assign would raise (documented generalisation,
    SURVEY.md section 8a A9); identical whenever ptr+n <= Q."""
    q = bank.shape[0]
    idx = (torch.arange(rows.shape[0]) + ptr) % q
    bank[idx] = rows


# --------------------------------------------------------------------------- #
# the loss block (train.py:191-266), differentiable w.r.t. logits / feats
# --------------------------------------------------------------------------- #
def loss_block(z_s, f_s, z_w, f_w, Y, bt: int, bank_feats, bank_probs,
               smooth: bool, adap_mask: float, hp: HyperParams):
    """z_*: logits [n,K]; f_*: l2-normalised feats [n,1024] of Base (s) / Base1 (w);
    rows [:bt] labelled, [bt:] unlabelled.  Returns a dict with the two total
    losses (autograd-connected) and every intermediate the tests compare."""
    K = z_s.shape[1]
    T = hp.temperature
    zL_s, zU_s = z_s[:bt], z_s[bt:]
    zL_w, zU_w = z_w[:bt], z_w[bt:]
    fL_s, fU_s = f_s[:bt], f_s[bt:]
    fL_w, fU_w = f_w[:bt], f_w[bt:]
    btu = zU_s.shape[0]

    cls_s = F.cross_entropy(zL_s, Y)                                        # :191
    cls_w = F.cross_entropy(zL_w, Y)                                        # :192
    pred_w = zL_w.argmax(1)                                                 # :194
    acc = (pred_w == Y).float().mean()                                      # :278

    with torch.no_grad():
        p_w = torch.softmax(zU_w.detach(), dim=1)                           # :203  ("probs")
        p_w0 = p_w.clone()                                                  # :205
        p_s = torch.softmax(zU_s.detach(), dim=1)                           # :209  ("probs1")
        p_s0 = p_s.clone()                                                  # :210
        if smooth:                                                          # :212
            A = torch.exp(torch.mm(fU_w.detach(), bank_feats[0].t()) / T)   # :213
            A = A / A.sum(1, keepdim=True)                                  # :214
            p_w = hp.alpha * p_w + (1 - hp.alpha) * torch.mm(A, bank_probs[0])   # :215
            A1 = torch.exp(torch.mm(fU_s.detach(), bank_feats[1].t()) / T)  # :217
            A1 = A1 / A1.sum(1, keepdim=True)                               # :218
            p_s = hp.alpha * p_s + (1 - hp.alpha) * torch.mm(A1, bank_probs[1])  # :219
        mask_w = p_w.max(1)[0].ge(adap_mask).float()                        # :220-222 ("mask")
        mask_s = p_s.max(1)[0].ge(adap_mask).float()                        # :227-228 ("masks")
        onehot = torch.zeros(bt, K).scatter(1, Y.view(-1, 1), 1)            # :224
        # rows written to the banks (:223,225,229,230): unlabelled feats of one net
        # stacked on LABELLED feats of the OTHER net
        bank0_rows = (torch.cat([fU_w.detach(), fL_s.detach()], 0), torch.cat([p_w0, onehot], 0))
        bank1_rows = (torch.cat([fU_s.detach(), fL_w.detach()], 0), torch.cat([p_s0, onehot], 0))

    con_s = (-(F.log_softmax(zU_s, dim=1) * p_w).sum(1) * mask_w).mean()    # :239,241
    con_w = (-(F.log_softmax(zU_w, dim=1) * p_s).sum(1) * mask_s).mean()    # :240,242

    sim = torch.exp(torch.mm(fU_s, fU_w.detach().t()) / T)                  # :246
    P = sim / sim.sum(1, keepdim=True)                                      # :247
    with torch.no_grad():
        Q0 = torch.mm(p_s, p_w.t())                                         # :249
        Q0.fill_diagonal_(1)                                                # :250
        pos_mask = (Q0 >= hp.pos_thr).float()                               # :251
        Q = Q0 * pos_mask                                                   # :252
        Q = Q / Q.sum(1, keepdim=True)                                      # :253
        neg_mask = (Q0 <= hp.neg_thr).float()                               # :254
        Qn = (1 - Q0) * neg_mask                                            # :255
        Qn = Qn / (Qn.sum(1, keepdim=True) + 1e-8)                          # :256
    sim1 = torch.exp(torch.mm(fU_s.detach(), fU_w.t()) / T)                 # :257
    P1 = sim1 / sim1.sum(1, keepdim=True)                                   # :258

    ctr_s = (-(torch.log(P) * Q).sum(1)).mean() + ((torch.log(P + 1) * Qn).sum(1)).mean()     # :260-262
    ctr_w = (-(torch.log(P1) * Q).sum(1)).mean() + ((torch.log(P1 + 1) * Qn).sum(1)).mean()   # :263-265

    total_s = cls_s + hp.w_contrast * ctr_s + hp.w_mutual * con_s           # :266
    total_w = cls_w + hp.w_contrast * ctr_w + hp.w_mutual * con_w           # :270
    return dict(total_s=total_s, total_w=total_w, cls_s=cls_s, cls_w=cls_w,
                con_s=con_s, con_w=con_w, ctr_s=ctr_s, ctr_w=ctr_w, acc=acc,
                p_w=p_w, p_s=p_s, p_w0=p_w0, p_s0=p_s0, mask_w=mask_w, mask_s=mask_s,
                Q=Q, Qn=Qn, P=P.detach(), bank0_rows=bank0_rows, bank1_rows=bank1_rows,
                n_pos=float(pos_mask.sum()), n_neg=float(neg_mask.sum()))


# --------------------------------------------------------------------------- #
# one training step  (train.py:150-278)
# --------------------------------------------------------------------------- #
def train_step(state: StepState, XPl, Xl, Y, XPu, Xu, noise: Sequence[torch.Tensor],
               dropmask: Sequence[Optional[torch.Tensor]], epoch: int, batch_index: int,
               hp: Optional[HyperParams] = None, apply_update: bool = True,
               relu_gates: Optional[Sequence[Optional[dict]]] = None):
    """``noise``: the 8 N(0,1) draws in reference order
        [XPl->Base, Xl->Base, XPl->Base1, Xl->Base1, XPu->Base, Xu->Base, XPu->Base1, Xu->Base1]
    (train.py:157,158,163,164,170,171,181,182).  ``dropmask``: per-network
    explicit dropout multiplier [n, cls_in] (or None).  ``relu_gates``: per-network
    activation patterns for basenet2_forward (tests only).  Mutates ``state``."""
    hp = hp or HyperParams()
    bt = XPl.shape[0]
    btu = XPu.shape[0]
    n = bt + btu
    sg = hp.noise
    XP_b_all = torch.cat([XPl + noise[0] * sg, XPu + noise[4] * sg], 0)     # :157,170,173
    X_b_all = torch.cat([Xl + noise[1] * sg, Xu + noise[5] * sg], 0)        # :158,171,174
    XP_e_all = torch.cat([XPl + noise[2] * sg, XPu + noise[6] * sg], 0)     # :163,181,183
    X_e_all = torch.cat([Xl + noise[3] * sg, Xu + noise[7] * sg], 0)        # :164,182,184

    ps = []
    for net in range(2):
        ps.append({k: (v.detach().clone().requires_grad_(True) if k in LIVE_KEYS else v)
                   for k, v in state.params[net].items()})
    taps = [{}, {}]
    rg = relu_gates or (None, None)
    z_s, f_s = basenet2_forward(ps[0], XP_b_all, X_b_all, dropmask[0], taps[0], rg[0])   # :175
    z_w, f_w = basenet2_forward(ps[1], XP_e_all, X_e_all, dropmask[1], taps[1], rg[1])   # :185

    smooth = (epoch > 0) or (batch_index > hp.queue_batch)                  # :212
    adap_mask = hp.thr * hp.adap_thr(epoch)                                 # :221
    lb = loss_block(z_s, f_s, z_w, f_w, Y, bt, state.bank_feats, state.bank_probs,
                    smooth, adap_mask, hp)

    # bank update happens after the smoothing read (:232-237)
    bank_write(state.bank_feats[0], state.ptr[0], lb["bank0_rows"][0])      # :232
    bank_write(state.bank_probs[0], state.ptr[0], lb["bank0_rows"][1])      # :233
    q = state.bank_feats[0].shape[0]
    new_ptr0 = (state.ptr[0] + hp.bank_step) % q                            # :234
    bank_write(state.bank_feats[1], state.ptr[1], lb["bank1_rows"][0])      # :235
    bank_write(state.bank_probs[1], state.ptr[1], lb["bank1_rows"][1])      # :236
    new_ptr1 = (new_ptr0 + hp.bank_step) % q                                # :237 (uses queue_ptr!)
    state.ptr = [new_ptr0, new_ptr1]

    g_s = torch.autograd.grad(lb["total_s"], [ps[0][k] for k in LIVE_KEYS])    # :267
    g_w = torch.autograd.grad(lb["total_w"], [ps[1][k] for k in LIVE_KEYS])    # :271
    grads = [dict(zip(LIVE_KEYS, g_s)), dict(zip(LIVE_KEYS, g_w))]
    if apply_update:
        for net in range(2):                                                # :268,272
            st = state.adam[net]
            st.t += 1
            for k in LIVE_KEYS:
                adam_update(state.params[net][k], grads[net][k], st.m[k], st.v[k], st.t, hp)

    out = dict(lb)
    out.update(logits=[z_s.detach(), z_w.detach()], feats=[f_s.detach(), f_w.detach()],
               grads=grads, smooth=smooth, adap_mask=adap_mask, n=n, taps=taps,
               # loss_hist row, train.py:274-278
               hist=[float(lb[k].detach()) for k in ("ctr_s", "total_s", "cls_s", "con_s", "acc")])
    return out


# --------------------------------------------------------------------------- #
# deterministic synthetic batches (shared by fixtures, tests, bench)
# --------------------------------------------------------------------------- #
def synthetic_batch(shape: NetShape, bt: int, btu: int, seed: int, with_noise: bool = True,
                    dropout: float = 0.8, separable: float = 0.0):
    """Closed-form batch from numpy PCG64: XP, X ~ N(0,1) (real data is z-scored,
    tools/hyper_tools.py:289-292), Y ~ U{0..K-1}, 8 noise draws, 2 dropout masks.
    ``separable`` > 0 adds a class-dependent offset so predictions become peaky."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    XPl = rng.standard_normal((bt, shape.C, shape.H, shape.W))
    Xl = rng.standard_normal((bt, shape.bands))
    Y = rng.integers(0, shape.K, size=(bt,))
    XPu = rng.standard_normal((btu, shape.C, shape.H, shape.W))
    Xu = rng.standard_normal((btu, shape.bands))
    if separable > 0:
        Yu = rng.integers(0, shape.K, size=(btu,))
        proto_p = rng.standard_normal((shape.K, shape.C, 1, 1)) * separable
        proto_x = rng.standard_normal((shape.K, shape.bands)) * separable
        XPl = XPl + proto_p[Y]; Xl = Xl + proto_x[Y]
        XPu = XPu + proto_p[Yu]; Xu = Xu + proto_x[Yu]
    out = dict(XPl=f32(XPl), Xl=f32(Xl), Y=torch.from_numpy(Y.astype(np.int64)),
               XPu=f32(XPu), Xu=f32(Xu))
    if with_noise:
        shp = [XPl.shape, Xl.shape, XPl.shape, Xl.shape, XPu.shape, Xu.shape, XPu.shape, Xu.shape]
        out["noise"] = [f32(rng.standard_normal(s)) for s in shp]
        n = bt + btu
        if dropout > 0:
            keep = 1.0 - dropout
            out["dropmask"] = [f32((rng.random((n, shape.cls_in)) < keep) / keep) for _ in range(2)]
        else:
            out["dropmask"] = [None, None]
    return out


def kill_spectral_rows(params: Sequence[Dict[str, torch.Tensor]], batch: dict, lab_row: int, unl_row: int) -> None:
    """Input generator for the dead-ReLU regime (tools/models.py:87-90 divides by a zero
    norm): make feat_spe's bias non-positive in every given parameter set and zero one labelled and one unlabelled
    spectrum together with their noise draws, so that exactly those rows leave the spectral ReLU all-zero and
    Normalize returns 0/0 = NaN for them.  In place."""
    for p in params:
        p["feat_spe.bias"] = -p["feat_spe.bias"].abs()
    if lab_row >= 0:
        batch["Xl"][lab_row] = 0
        batch["noise"][1][lab_row] = 0
        batch["noise"][3][lab_row] = 0
    if unl_row >= 0:
        batch["Xu"][unl_row] = 0
        batch["noise"][5][unl_row] = 0
        batch["noise"][7][unl_row] = 0


# --------------------------------------------------------------------------- #
# next row N3: patch extraction  (tools/hyper_tools.py:35-55 MirrowCut, :226-243 ExtractPatches)
# --------------------------------------------------------------------------- #
def mirror_index(t: np.ndarray, size: int) -> np.ndarray:
    """Index into the original axis of the symmetric (edge-repeating) mirror extension built by
    MirrowCut (hyper_tools.py:35-55): t < 0 -> -t-1 ; t >= size -> 2*size-1-t."""
    t = np.asarray(t)
    return np.where(t < 0, -t - 1, np.where(t >= size, 2 * size - 1 - t, t))


def extract_patches(X: np.ndarray, w: int, pixel_idx: Optional[np.ndarray] = None) -> np.ndarray:
    """X: [row, col, C] cube -> patches [n, C, w, w] for the given row-major pixel indices (all pixels
    when None).  Restates ExtractPatches (hyper_tools.py:226-243): pixel (r, c) takes mirror-extended rows
    r-hw .. r-hw+w-1 and columns c-hw .. c-hw+w-1 with hw = w // 2, then moveaxis(3, 1).  For even w this
    is exactly the reference (its slice [index-hw : index+hw] has 2*hw = w rows); for odd w the reference
    raises a broadcast error, and this is the centred generalisation."""
    row, col, C = X.shape
    hw = w // 2
    if pixel_idx is None:
        pixel_idx = np.arange(row * col)
    pixel_idx = np.asarray(pixel_idx, dtype=np.int64)
    r, c = pixel_idx // col, pixel_idx % col
    off = np.arange(w) - hw
    rr = mirror_index(r[:, None] + off[None, :], row)          # [n, w]
    cc = mirror_index(c[:, None] + off[None, :], col)          # [n, w]
    patches = X[rr[:, :, None], cc[:, None, :], :]             # [n, w, w, C]
    return np.ascontiguousarray(np.moveaxis(patches, 3, 1)).astype(np.float32)


# --------------------------------------------------------------------------- #
# next row N4: tools.models.ContrastiveLoss  (tools/models.py:14-39, NT-Xent)
# --------------------------------------------------------------------------- #
def ntxent_loss(emb_i: torch.Tensor, emb_j: torch.Tensor, temperature: float = 0.5) -> torch.Tensor:
    """Restates ContrastiveLoss.forward (models.py:22-39): rows are L2-normalised (F.normalize, eps 1e-12),
    all-pairs cosine similarity of the 2B representations, positives on the +-B diagonals, denominator
    over every k != i."""
    B = emb_i.shape[0]
    z = torch.cat([F.normalize(emb_i, dim=1), F.normalize(emb_j, dim=1)], dim=0)            # :23-26
    sim = F.cosine_similarity(z.unsqueeze(1), z.unsqueeze(0), dim=2)                         # :27
    pos = torch.cat([torch.diag(sim, B), torch.diag(sim, -B)], dim=0)                        # :30-32
    neg_mask = (~torch.eye(2 * B, dtype=torch.bool)).float()                                  # :19-20
    denom = (neg_mask * torch.exp(sim / temperature)).sum(dim=1)                              # :35
    return (-torch.log(torch.exp(pos / temperature) / denom)).sum() / (2 * B)                 # :34,37-38


# --------------------------------------------------------------------------- #
# next row N2: loss_helper.py  (memory-bank InfoNCE + entropy-filtered CE)
# --------------------------------------------------------------------------- #
MB_DELTA_P = 0.3        # loss_helper.py:56  current_class_threshold
MB_DELTA_N = 1.0        # :57  current_class_negative_threshold
MB_LOW_RANK, MB_HIGH_RANK = 3, 9   # :58
MB_TEMP = 0.5           # :59
MB_QUERIES = 256        # :60
MB_NEGATIVES = 50       # :61
IGNORE = 255            # :229,253


def memobank_enqueue(keys: torch.Tensor, queue: torch.Tensor, ptr: int, size: int):
    """dequeue_and_enqueue (loss_helper.py:19-36) as a pure function: the bank of one class is a FIFO of at most
    `size` rows -- append the new keys, keep the LAST `size` rows.  The pointer becomes `size` once the bank is
    full, (ptr + m) % size before that.  Returns (queue, ptr, m)."""
    m = int(keys.shape[0])
    joined = torch.cat((queue, keys.detach().clone()), dim=0)
    if joined.shape[0] >= size:
        return joined[joined.shape[0] - size:], size, m
    return joined, (int(ptr) + m) % size, m


def class_ranks(prob: torch.Tensor) -> torch.Tensor:
    """rank[n, c] = position of class c when row n is sorted by descending probability (what
    torch.sort(prob, 1, True)[1] encodes, loss_helper.py:77,81); ties broken by class index."""
    gt = (prob.unsqueeze(1) > prob.unsqueeze(2)).sum(dim=2)                       # [n, c]: #{j: p_j > p_c}
    K = prob.shape[1]
    earlier = torch.tril(torch.ones(K, K, dtype=torch.bool), diagonal=-1)         # [c, j]: j < c
    eq = ((prob.unsqueeze(1) == prob.unsqueeze(2)) & earlier.unsqueeze(0)).sum(dim=2)
    return gt + eq


def percentile_threshold(values: np.ndarray, percent: float):
    """np.percentile(values, percent) (loss_helper.py:250-252), default 'linear' method."""
    return np.percentile(values, percent)


def unsupervised_loss(predict: torch.Tensor, target: torch.Tensor, percent: float, pred_teacher: torch.Tensor):
    """compute_unsupervised_loss (loss_helper.py:242-261).  Rows whose teacher entropy is at or above the
    `percent`-th percentile of the valid rows' entropies are set to IGNORE; the loss is
    (B / #kept) * mean-over-kept CE(predict, target).  Returns (loss, target_after) -- the reference edits
    `target` in place."""
    B = predict.shape[0]
    with torch.no_grad():
        p = torch.softmax(pred_teacher, dim=1)
        entropy = -(p * torch.log(p + 1e-10)).sum(dim=1)                          # :247-248
        valid = target != IGNORE
        thresh = percentile_threshold(entropy[valid].numpy().flatten(), percent)  # :250-252
        drop = (entropy >= float(thresh)) & valid                                 # :253
        tgt = torch.where(drop, torch.full_like(target, IGNORE), target)          # :255
        kept = tgt != IGNORE
        weight = B / kept.sum()                                                   # :256
    logp = F.log_softmax(predict, dim=1)
    picked = logp.gather(1, torch.where(kept, tgt, torch.zeros_like(tgt)).unsqueeze(1)).squeeze(1)
    ce = -(picked * kept.float()).sum() / kept.sum()                              # :258 (ignore_index mean)
    return weight * ce, tgt


def contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, memobank, ptrs, sizes,
                         rep_teacher, anchor_idx=None, neg_idx=None, momentum_prototype=None, i_iter=0):
    """compute_contra_memobank_loss (loss_helper.py:39-219) with the random draws injected.

    rep, rep_teacher [N, D]; label_* one-hot [*, K]; prob_* [*, K]; low_mask / high_mask [N, 1];
    memobank: list (per class) of [m_c, D] tensors; ptrs / sizes: lists of ints.
    anchor_idx[c] int64 [256]: positions into class c's low-entropy list (the reference's first randint, :164);
    neg_idx[c] int64 [256*50]: rows of the class's bank AFTER this call's enqueue (second randint, :179).
    Returns dict(loss, new_keys, memobank, ptrs, valid_classes, prototype).

    Kept quirk (:158-190): the loop over the `valid_seg` valid classes indexes the per-class anchor lists and
    prototypes by POSITION i (0..valid_seg-1), but the bank by valid_classes[i]."""
    K = label_l.shape[1]
    Nl = label_l.shape[0]
    D = rep.shape[1]
    label = torch.cat((label_l, label_u), dim=0)
    low_valid = label * low_mask                                                   # :67
    high_valid = label * high_mask                                                 # :68
    prob = torch.cat((prob_l, prob_u), dim=0)                                      # :86
    rank_l, rank_u = class_ranks(prob_l), class_ranks(prob_u)
    anchors_pool, protos, new_keys, valid_classes, counts = [], [], [], [], []
    bank = [b.clone() for b in memobank]
    ptr_out = list(ptrs)
    for c in range(K):
        lv = low_valid[:, c].bool()
        hv = high_valid[:, c].bool()
        low_entropy = (prob[:, c] > MB_DELTA_P) & lv                               # :95
        high_entropy = (prob[:, c] < MB_DELTA_N) & hv                              # :96
        anchors_pool.append(rep[low_entropy])                                      # :99
        protos.append(rep_teacher[lv].detach().mean(dim=0, keepdim=True))          # :102-106 (NaN row if empty)
        in_u = (rank_u[:, c] >= MB_LOW_RANK) & (rank_u[:, c] < MB_HIGH_RANK)       # :110-112
        in_l = (rank_l[:, c] < MB_LOW_RANK) & (label_l[:, c] == 0)                 # :117-121
        negative = high_entropy & torch.cat((in_l, in_u), dim=0)                   # :123
        bank[c], ptr_out[c], m = memobank_enqueue(rep_teacher[negative].detach(), bank[c], ptr_out[c], sizes[c])
        new_keys.append(m)                                                         # :126-133
        if int(lv.sum()) > 0:                                                      # :135-137
            counts.append(int(lv.sum()))
            valid_classes.append(c)
    out = dict(new_keys=new_keys, memobank=bank, ptrs=ptr_out, valid_classes=valid_classes, prototype=None)
    if len(counts) <= 1:                                                           # :139-145
        out["loss"] = 0.0 * rep.sum()
        return out
    proto = torch.cat(protos, dim=0)                                               # :149  [K, D]
    valid_seg = len(counts)
    prototype = torch.zeros(K, MB_QUERIES, 1, D)
    total = torch.zeros(())
    for i in range(valid_seg):
        vc = valid_classes[i]
        if anchors_pool[i].shape[0] == 0 or bank[vc].shape[0] == 0:                # :158-172
            total = total + 0.0 * rep.sum()
            continue
        anchor = anchors_pool[i][anchor_idx[i]]                                    # :164-168  [Q, D]
        with torch.no_grad():
            negs = bank[vc][neg_idx[i]].reshape(MB_QUERIES, MB_NEGATIVES, D)       # :176-185
            pos = proto[i].view(1, 1, D).repeat(MB_QUERIES, 1, 1)                  # :186-192
            if momentum_prototype is not None:                                     # :194-203
                if not bool((momentum_prototype == 0).all()):
                    ema = min(1 - 1 / i_iter, 0.999)
                    pos = (1 - ema) * pos + ema * momentum_prototype[vc]
                prototype[vc] = pos.clone()
            keys = torch.cat((pos, negs), dim=1)                                   # :205-207  [Q, 51, D]
        logits = torch.cosine_similarity(anchor.unsqueeze(1), keys, dim=2)         # :209-211
        total = total + F.cross_entropy(logits / MB_TEMP, torch.zeros(MB_QUERIES, dtype=torch.long))  # :213-215
    out["loss"] = total / valid_seg                                                # :217-219
    if momentum_prototype is not None:
        out["prototype"] = prototype
    return out


def contra_draw_plan(label_l, label_u, prob_l, prob_u, low_mask, high_mask, memobank, sizes, rep_teacher_rows=None):
    """Which loop positions of compute_contra_memobank_loss draw random indices, and from what ranges
    (loss_helper.py:158-183): a list of (position i, anchor pool size, bank rows of valid_classes[i] after this
    call's enqueue); empty when at most one class is valid."""
    K = label_l.shape[1]
    label = torch.cat((label_l, label_u), dim=0)
    prob = torch.cat((prob_l, prob_u), dim=0)
    low_valid = label * low_mask
    high_valid = label * high_mask
    rank_l, rank_u = class_ranks(prob_l), class_ranks(prob_u)
    pool, rows, valid = [], [], []
    for c in range(K):
        lv = low_valid[:, c].bool()
        pool.append(int(((prob[:, c] > MB_DELTA_P) & lv).sum()))
        in_u = (rank_u[:, c] >= MB_LOW_RANK) & (rank_u[:, c] < MB_HIGH_RANK)
        in_l = (rank_l[:, c] < MB_LOW_RANK) & (label_l[:, c] == 0)
        neg = ((prob[:, c] < MB_DELTA_N) & high_valid[:, c].bool()) & torch.cat((in_l, in_u), dim=0)
        rows.append(min(int(memobank[c].shape[0]) + int(neg.sum()), int(sizes[c])))
        if int(lv.sum()) > 0:
            valid.append(c)
    if len(valid) <= 1:
        return []
    return [(i, pool[i], rows[valid[i]]) for i in range(len(valid)) if pool[i] > 0 and rows[valid[i]] > 0]

