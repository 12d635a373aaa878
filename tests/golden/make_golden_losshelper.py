#!/usr/bin/env python3
"""Golden vectors for loss_helper.py: run the reference's own functions on seeded inputs.

  * loads /root/reference/loss_helper.py BY PATH; `.cuda()` is patched to identity (CPU run);
  * `torch.randint` inside the module is served by a proxy that draws from a seeded numpy generator and records
    the (range, count) of every draw, so tests regenerate the same indices from the seed (tests/losshelper_util.py)
    and inject them into the oracle / the HIP path;
  * stores inputs-by-seed, the recorded indices and OUTPUT vectors only.
Build container only:  python tests/golden/make_golden_losshelper.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden._refload import load_reference_module  # noqa: E402
from tests.losshelper_util import CASES_CONTRA, CASES_UNSUP, GRAD_ROWS, contra_inputs, unsup_inputs  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
ref = load_reference_module("loss_helper.py", "ref_loss_helper")


class TorchProxy:
    def __init__(self, seed):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.highs = []          # (range, count) of every draw: enough to regenerate it from the seed

    def randint(self, high, size):
        t = torch.from_numpy(self.rng.integers(0, int(high), size=tuple(size), dtype=np.int64))
        self.highs.append((int(high), int(np.prod(size))))
        return t

    def __getattr__(self, name):
        return getattr(torch, name)


out = {}
for name, cfg in CASES_UNSUP.items():
    predict, target, teacher = unsup_inputs(cfg)
    predict.requires_grad_(True)
    tgt = target.clone()
    loss = ref.compute_unsupervised_loss(predict, tgt, cfg["percent"], teacher)
    loss.backward()
    out[f"u_{name}_loss"] = np.array([loss.item()])
    out[f"u_{name}_target"] = tgt.numpy().astype(np.int16) if cfg["B"] > 2048 else tgt.numpy()
    out[f"u_{name}_grad"] = predict.grad.numpy()[:GRAD_ROWS] if cfg["B"] > 2048 else predict.grad.numpy()
    out[f"u_{name}_gnorm"] = np.array([np.sqrt((predict.grad.double().numpy() ** 2).sum())])

for name, cfg in CASES_CONTRA.items():
    inp = contra_inputs(cfg)
    K, D = cfg["K"], cfg["D"]
    rep = inp["rep"].clone().requires_grad_(True)
    memobank = [[b.clone()] for b in inp["bank"]]
    ptrs = [torch.tensor([p], dtype=torch.long) for p in inp["ptrs"]]
    proxy = TorchProxy(cfg["seed"] + 1000)
    ref.torch = proxy
    mp = inp.get("momentum")
    res = ref.compute_contra_memobank_loss(rep, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"],
                                           inp["low_mask"], inp["high_mask"], memobank, ptrs, list(inp["sizes"]),
                                           inp["rep_teacher"], momentum_prototype=mp, i_iter=cfg.get("i_iter", 0))
    ref.torch = torch
    if mp is None:
        new_keys, loss = res
    else:
        prototype, new_keys, loss = res
        out[f"c_{name}_prototype_sum"] = np.array([prototype.double().sum().item()])
    loss.backward()
    out[f"c_{name}_loss"] = np.array([loss.item()])
    out[f"c_{name}_new_keys"] = np.array(new_keys, dtype=np.int64)
    out[f"c_{name}_ptrs"] = np.array([int(p[0]) for p in ptrs], dtype=np.int64)
    out[f"c_{name}_bank_rows"] = np.array([memobank[c][0].shape[0] for c in range(K)], dtype=np.int64)
    out[f"c_{name}_bank_sum"] = np.array([memobank[c][0].double().sum().item() for c in range(K)])
    out[f"c_{name}_bank_last"] = np.stack([memobank[c][0][-1, :4].numpy() if memobank[c][0].shape[0] else
                                           np.zeros(4, np.float32) for c in range(K)])
    g = rep.grad.numpy() if rep.grad is not None else np.zeros((cfg["Nl"] + cfg["Nu"], D), np.float32)
    out[f"c_{name}_grad"] = g[:, :16].copy()                      # first 16 feature columns + the full norm
    out[f"c_{name}_gnorm"] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())])
    out[f"c_{name}_draws"] = np.array(proxy.highs, dtype=np.int64).reshape(-1, 2)

np.savez_compressed(os.path.join(HERE, "losshelper_ref.npz"), **out)
for k, v in out.items():
    if v.size <= 20:
        print(k, v.tolist())
print("size", os.path.getsize(os.path.join(HERE, "losshelper_ref.npz")) // 1024, "KB")
