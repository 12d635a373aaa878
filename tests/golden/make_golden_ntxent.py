#!/usr/bin/env python3
"""Golden vectors for tools.models.ContrastiveLoss: run the reference's own class
(tools/models.py:14-39, device='cpu') on seeded embeddings; store loss + gradient slices.
Build container only:  python tests/golden/make_golden_ntxent.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden._refload import load_reference_module  # noqa: E402

ref = load_reference_module("tools/models.py", "ref_models")
out = {}
for name, (B, D, T, seed) in {"a": (8, 16, 0.5, 61), "b": (32, 128, 0.3, 62), "c": (64, 1024, 0.5, 63)}.items():
    rng = np.random.Generator(np.random.PCG64(seed))
    ei = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).requires_grad_(True)
    ej = torch.from_numpy((rng.standard_normal((B, D)) * 2 + 0.3).astype(np.float32)).requires_grad_(True)
    loss = ref.ContrastiveLoss(B, device="cpu", temperature=T)(ei, ej)
    loss.backward()
    out[name + "_cfg"] = np.array([B, D, seed], dtype=np.int64)
    out[name + "_T"] = np.array([T])
    out[name + "_loss"] = np.array([loss.item()])
    out[name + "_gi"] = ei.grad.numpy()[:, :8].copy()
    out[name + "_gj"] = ej.grad.numpy()[:, :8].copy()
    out[name + "_gnorm"] = np.array([ei.grad.double().norm().item(), ej.grad.double().norm().item()])
np.savez_compressed(os.path.join(HERE, "ntxent_ref.npz"), **out)
print({k: (v.tolist() if v.size < 4 else v.shape) for k, v in out.items()})
