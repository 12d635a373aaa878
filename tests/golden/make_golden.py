#!/usr/bin/env python3
"""Generate golden fixtures by running THE REFERENCE ITSELF on CPU.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

What it does:
  * inserts an empty ``torchvision`` module (tools/models.py:6 imports it, unused),
    imports the reference ``tools.models.BaseNet2``;
  * reads ``train.py`` as text, takes lines 150-279 (the inline training step),
    dedents and exec's them in a namespace supplying the surrounding locals of
    ``main()``; ``.cuda()`` is patched to identity; ``torch.randn`` is served
    from a pre-generated noise list and ``net.drop`` is swapped for an explicit
    mask multiply so the run is reproducible from numpy seeds;
  * for shapes the reference cannot construct (conv0 hard-codes 60 channels,
    classifier 2624 inputs) it swaps ``conv0``/``classifier`` post-construction
    (forward code is shape-agnostic).
Only inputs-by-seed and OUTPUT VECTORS are stored (``*.npz``); no reference source.
"""
import os
import sys
import textwrap
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from tests.golden._refload import load_reference_module  # noqa: E402

RefBaseNet2 = load_reference_module("tools/models.py", "ref_models").BaseNet2     # the reference's own class

from oracle import cmlpl_oracle as O  # noqa: E402  (input generators only)

torch.set_num_threads(8)
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self

with open(os.path.join(REF, "train.py")) as fh:
    _lines = fh.read().split("\n")
STEP_SRC = textwrap.dedent("\n".join(_lines[149:279]))      # train.py:150-279
STEP_CODE = compile(STEP_SRC, "<reference train.py:150-279>", "exec")


class MaskDrop(nn.Module):
    """stands in for nn.Dropout (tools/models.py:116): explicit multiplier."""
    def __init__(self):
        super().__init__()
        self.mask = None

    def forward(self, x):
        return x * self.mask if (self.training and self.mask is not None) else x


class TorchProxy:
    """``torch`` with randn() served from a list (CPU generator in the reference)."""
    def __init__(self):
        self.queue = []

    def randn(self, size):
        t = self.queue.pop(0)
        assert tuple(t.shape) == tuple(size), (t.shape, size)
        return t

    def __getattr__(self, name):
        return getattr(torch, name)


def build_ref_net(shape, params, dropout):
    net = RefBaseNet2(num_features=shape.bands, dropout=dropout, num_classes=shape.K)
    if shape.C != 60:
        net.conv0 = nn.Conv2d(shape.C, 64, kernel_size=1, stride=1, bias=True)
    if shape.cls_in != 2624:
        net.classifier = nn.Linear(shape.cls_in, shape.K)
    net.load_state_dict(params)
    net.drop = MaskDrop()
    return net


LIVE = O.LIVE_KEYS


def run_case(name, shape, bt, btu, steps, seed, epoch0=0, batch_index0=0, dropout=0.8,
             cls_scale=1.0, separable=0.0, full_steps=(0,), num_epochs=20, thr=1.0, dead=(-1, -1)):
    hp = O.HyperParams(num_epochs=num_epochs, thr=thr, dropout=dropout)
    args = types.SimpleNamespace(noise=hp.noise, queue_batch=hp.queue_batch, temperature=hp.temperature,
                                 alpha=hp.alpha, thr=hp.thr, labeled_batch_size=bt,
                                 unlabeled_batch_size=btu, lr=hp.lr, num_epochs=num_epochs)
    p0 = O.closed_form_params(shape, seed)
    p1 = O.closed_form_params(shape, seed + 1)
    if cls_scale != 1.0:
        p0["classifier.weight"] *= cls_scale
        p1["classifier.weight"] *= cls_scale
    if dead != (-1, -1):      # dead-ReLU regime: bias made non-positive here, the rows zeroed per batch below
        O.kill_spectral_rows([p0, p1], dict(Xl=torch.zeros(bt, 1), Xu=torch.zeros(btu, 1),
                                            noise=[torch.zeros(max(bt, btu), 1)] * 8), -1, -1)
    Base = build_ref_net(shape, p0, dropout)
    Base1 = build_ref_net(shape, p1, dropout)
    tp = TorchProxy()
    queue_size = 5 * bt * 2
    ns = dict(args=args, Base=Base, Base1=Base1, cls_loss=torch.nn.CrossEntropyLoss(),
              base_optimizer=torch.optim.Adam(Base.parameters(), lr=args.lr),
              base1_optimizer=torch.optim.Adam(Base1.parameters(), lr=args.lr),
              loss_hist=np.zeros((steps, 5)), index_i=-1,
              queue_size=queue_size, queue_size1=queue_size,
              queue_feats=torch.zeros(queue_size, 1024), queue_probs=torch.zeros(queue_size, shape.K),
              queue_ptr=0,
              queue_feats1=torch.zeros(queue_size, 1024), queue_probs1=torch.zeros(queue_size, shape.K),
              queue_ptr1=0, num_classes=shape.K, print_per_batches=10, num_batches=10 ** 9,
              torch=tp, np=np, F=F, time=__import__("time"))
    rec = {k: [] for k in ("hist", "extra", "ptr", "counts", "logit_sums", "grad_norms",
                           "param_sums", "bank_sums", "grad_nan", "bank_nan")}
    full = {}
    for s in range(steps):
        epoch = epoch0
        batch_index = batch_index0 + s
        b = O.synthetic_batch(shape, bt, btu, seed * 1000 + s, dropout=dropout, separable=separable)
        if dead != (-1, -1):
            O.kill_spectral_rows([], b, dead[0], dead[1])
        tp.queue = list(b["noise"])
        Base.drop.mask, Base1.drop.mask = b["dropmask"]
        ns.update(epoch=epoch, batch_index=batch_index,
                  adap_thr=np.exp(-0.5 * ((epoch / num_epochs) ** 2)),       # train.py:147-148
                  labeled_data=(b["XPl"], b["Xl"], b["Y"]),
                  unlabeled_data=(b["XPu"], b["Xu"], torch.zeros(btu, dtype=torch.long)))
        exec(STEP_CODE, ns)
        assert not tp.queue
        rec["hist"].append(ns["loss_hist"][ns["index_i"]].copy())
        rec["extra"].append([ns["total_loss1"].item(), ns["cls_loss_value1"].item(),
                             ns["con_loss_value1"].item(), ns["loss_contrast1"].item()])
        rec["ptr"].append([ns["queue_ptr"], ns["queue_ptr1"]])
        rec["counts"].append([ns["mask"].sum().item(), ns["masks"].sum().item(),
                              ns["pos_mask"].sum().item(), ns["neg_mask"].sum().item()])
        rec["logit_sums"].append([ns["un_b_output_all"].sum().item(), ns["un_b_output_all"].abs().sum().item(),
                                  ns["un_e_output_all"].sum().item(), ns["un_e_output_all"].abs().sum().item()])
        gn, psum, gnan = [], [], []
        for net in (Base, Base1):
            sd = dict(net.named_parameters())
            gn.append([sd[k].grad.double().norm().item() for k in LIVE])
            psum.append([sd[k].detach().double().sum().item() for k in LIVE])
            gnan.append([int(torch.isnan(sd[k].grad).sum()) for k in LIVE])
        rec["grad_norms"].append(gn)
        rec["param_sums"].append(psum)
        rec["grad_nan"].append(gnan)          # where NaN lands (dead-ReLU regime), element counts
        rec["bank_nan"].append([int(torch.isnan(ns[k]).sum()) for k in
                                ("queue_feats", "queue_probs", "queue_feats1", "queue_probs1")])
        rec["bank_sums"].append([ns["queue_feats"].double().sum().item(), ns["queue_probs"].double().sum().item(),
                                 ns["queue_feats1"].double().sum().item(), ns["queue_probs1"].double().sum().item()])
        for net in (Base, Base1):
            for k, prm in net.named_parameters():
                if k not in LIVE:
                    assert prm.grad is None        # dead parameters
        if s in full_steps:
            full[f"s{s}_logits"] = np.stack([ns["un_b_output_all"].detach().numpy(),
                                             ns["un_e_output_all"].detach().numpy()])
            full[f"s{s}_feats8"] = np.stack([ns["xs_feature_all"].detach().numpy()[:, :8],
                                             ns["xw_feature_all"].detach().numpy()[:, :8]])
            full[f"s{s}_probs"] = np.stack([ns["probs"].numpy(), ns["probs1"].numpy()])
            full[f"s{s}_masks"] = np.stack([ns["mask"].numpy(), ns["masks"].numpy()])
            full[f"s{s}_Qdiag"] = np.stack([ns["Q"].diag().numpy(), ns["Q_n"].sum(1).numpy()])
            full[f"s{s}_grad_cls"] = np.stack([dict(Base.named_parameters())["classifier.weight"].grad.numpy()[:, :16],
                                               dict(Base1.named_parameters())["classifier.weight"].grad.numpy()[:, :16]])
    out = {k: np.asarray(v, dtype=np.float64) for k, v in rec.items()}
    out.update(full)
    out["cfg"] = np.asarray([shape.C, shape.H, shape.W, shape.bands, shape.K, bt, btu, steps, seed,
                             epoch0, batch_index0, num_epochs], dtype=np.int64)
    out["cfg_f"] = np.asarray([dropout, cls_scale, separable, thr], dtype=np.float64)
    out["full_steps"] = np.asarray(full_steps, dtype=np.int64)
    out["cfg_dead"] = np.asarray(dead, dtype=np.int64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {steps} steps, last hist={rec['hist'][-1]}, counts={rec['counts'][-1]}, "
          f"{os.path.getsize(path) / 1024:.1f} KB")


def main():
    P = O.NetShape(60, 20, 20, 103, 9)          # the only shape the reference constructs
    B2 = O.NetShape(103, 11, 11, 103, 9)        # BASELINE configs[1]
    B4 = O.NetShape(200, 11, 11, 200, 16)       # Indian-Pines-shaped
    B5 = O.NetShape(48, 15, 15, 48, 20)         # Houston2018-shaped
    # (i) random init, 25-step trajectory: smoothing gate flips at batch_index 18, bank wraps
    run_case("p_traj_32", P, 32, 32, steps=25, seed=11, full_steps=(0, 24))
    # reference-default batch, epoch 1 (smoothing from step 0), full bank wrap (5 steps) + 2
    run_case("p_b256_ep1", P, 128, 128, steps=7, seed=12, epoch0=1, full_steps=(0, 6))
    # (ii) peaky regime: thresholds, pos/neg masks and the mid band all fire
    run_case("p_peaky_32", P, 32, 32, steps=6, seed=13, epoch0=12, cls_scale=30.0, separable=1.5,
             full_steps=(0, 5), thr=0.9)
    # (iv) short batch, pointer still advances by the literal 256
    run_case("p_short_16", P, 16, 16, steps=6, seed=14, epoch0=1, full_steps=(5,))
    # dropout disabled path (self.dropout == 0 skips self.drop, models.py:147)
    run_case("p_nodrop_32", P, 32, 32, steps=3, seed=15, epoch0=1, dropout=0.0, full_steps=(2,))
    # BASELINE shapes via post-construction conv0/classifier swap
    run_case("b2_64", B2, 32, 32, steps=6, seed=21, epoch0=1, full_steps=(0, 5))
    run_case("b2_b256", B2, 128, 128, steps=3, seed=22, epoch0=1, full_steps=(2,))
    run_case("b4_64", B4, 32, 32, steps=3, seed=23, epoch0=1, full_steps=(2,))
    run_case("b5_64", B5, 32, 32, steps=3, seed=24, epoch0=1, full_steps=(2,))
    round2()


def round2(only=None):
    """Fixtures added in round 2 (the round-1 files are left byte-identical)."""
    B2 = O.NetShape(103, 11, 11, 103, 9)
    B5 = O.NetShape(48, 15, 15, 48, 20)
    cases = dict(
        # the headline configuration (B2, 128+128) in the regime where thresholds, pos/neg masks, the mid band and
        # the mutual loss all fire (con_s != 0)
        b2_peaky_256=lambda: run_case("b2_peaky_256", B2, 128, 128, steps=4, seed=31, epoch0=12, cls_scale=30.0,
                                      separable=1.5, full_steps=(0, 3), thr=0.9),
        # BASELINE configs[4]: 64 labelled + 512 unlabelled (1:8) on the 15x15x48 shape.  The reference's bank
        # slice-assign (train.py:232-236) only fits at ptr = 0: it runs exactly ONE step at this split and raises
        # at the next, so one step is all the reference can pin here.
        b5_1to8=lambda: run_case("b5_1to8", B5, 64, 512, steps=1, seed=32, epoch0=1, full_steps=(0,)),
        # (v) dead spectral ReLU rows: Normalize divides by a zero norm (tools/models.py:87-90) -> NaN features
        # for one labelled and one unlabelled row, propagated through sim / losses / gradients / banks
        b2_deadrelu_64=lambda: run_case("b2_deadrelu_64", B2, 32, 32, steps=2, seed=33, epoch0=1, full_steps=(0, 1),
                                        dead=(3, 5)),
    )
    for k, f in cases.items():
        if only is None or k in only:
            f()


def round3(only=None):
    """Fixtures added in round 3 (earlier files are left byte-identical)."""
    B4 = O.NetShape(200, 11, 11, 200, 16)
    cases = dict(
        # BASELINE configs[3] at its stated batch (Indian-Pines-shaped 11x11x200, 16 classes, 128 + 128 rows): the
        # shape that takes the chunked slab (two passes of the fused conv0 stages)
        b4_b256=lambda: run_case("b4_b256", B4, 128, 128, steps=3, seed=41, epoch0=1, full_steps=(0, 2)),
    )
    for k, f in cases.items():
        if only is None or k in only:
            f()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--round2":
        round2(sys.argv[2:] or None)
    elif len(sys.argv) > 1 and sys.argv[1] == "--round3":
        round3(sys.argv[2:] or None)
    else:
        main()
        round3()
