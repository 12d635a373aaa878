"""tools.models.ContrastiveLoss.  CPU: oracle restatement vs vectors produced by the
reference's own class.  GPU: HIP forward + analytic backward vs the oracle's autograd."""
import os

import numpy as np
import pytest
import torch

from oracle import cmlpl_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ntxent_ref.npz")


def _emb(B, D, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    ei = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
    ej = torch.from_numpy((rng.standard_normal((B, D)) * 2 + 0.3).astype(np.float32))
    return ei, ej


def test_oracle_matches_reference_contrastive_loss():
    z = np.load(GOLD)
    for name in ("a", "b", "c"):
        B, D, seed = (int(v) for v in z[name + "_cfg"])
        T = float(z[name + "_T"][0])
        ei, ej = _emb(B, D, seed)
        ei.requires_grad_(True); ej.requires_grad_(True)
        loss = O.ntxent_loss(ei, ej, T)
        loss.backward()
        assert abs(loss.item() - z[name + "_loss"][0]) <= 1e-6 * abs(z[name + "_loss"][0])
        assert np.allclose(ei.grad.numpy()[:, :8], z[name + "_gi"], rtol=1e-5, atol=1e-8)
        assert np.allclose(ej.grad.numpy()[:, :8], z[name + "_gj"], rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("B,D,T", [(8, 16, 0.5), (32, 128, 0.3), (64, 1024, 0.5), (128, 1024, 0.5), (20, 100, 0.2), (9, 37, 0.4),
                                   (150, 300, 0.5), (256, 1024, 0.5), (512, 1024, 0.5), (512, 200, 0.3), (300, 516, 0.4)])
def test_hip_contrastive_loss_matches_oracle(B, D, T):
    from tools.models import ContrastiveLoss
    from tests.memo import memo
    ei, ej = _emb(B, D, 70 + B)

    def oracle():          # (the reference's [2B, 2B, D] broadcast: 17 s at B = 512; the kernel variants share one evaluation)
        ri, rj = ei.clone().requires_grad_(True), ej.clone().requires_grad_(True)
        ref = O.ntxent_loss(ri, rj, T)
        ref.backward()
        return ref.detach(), ri.grad, rj.grad
    ref, ri_grad, rj_grad = memo(f"ntxent-{B}-{D}-{T}-seed{70 + B}", oracle)
    gi, gj = ei.cuda().requires_grad_(True), ej.cuda().requires_grad_(True)
    crit = ContrastiveLoss(B, device="cuda", temperature=T)
    loss = crit(gi, gj)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    for got, want in ((gi.grad, ri_grad), (gj.grad, rj_grad)):
        err = (got.cpu() - want).abs().max().item()
        assert err <= 2e-4 * want.abs().max().item() + 1e-9, (err, want.abs().max().item())
