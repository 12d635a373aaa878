"""loss_helper.py row: dequeue_and_enqueue, compute_unsupervised_loss,
compute_contra_memobank_loss.  CPU: the oracle restatement against vectors produced by the reference's own
functions (tests/golden/make_golden_losshelper.py, random draws recorded and injected).  GPU: the HIP path
through the C ABI against the oracle on the same inputs and draws."""
import os

import numpy as np
import pytest
import torch

from oracle import cmlpl_oracle as O
from tests.losshelper_util import CASES_CONTRA, CASES_UNSUP, contra_inputs, regenerate_draws, unsup_inputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losshelper_ref.npz")


def run_oracle_contra(name):
    cfg = CASES_CONTRA[name]
    inp = contra_inputs(cfg)
    rep = inp["rep"].clone().requires_grad_(True)
    plan = O.contra_draw_plan(inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"],
                              inp["high_mask"], inp["bank"], inp["sizes"])
    anchor_idx, neg_idx, highs = regenerate_draws(cfg, plan)
    res = O.contra_memobank_loss(rep, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"],
                                 inp["low_mask"], inp["high_mask"], inp["bank"], inp["ptrs"], inp["sizes"],
                                 inp["rep_teacher"], anchor_idx=anchor_idx, neg_idx=neg_idx,
                                 momentum_prototype=inp.get("momentum"), i_iter=cfg.get("i_iter", 0))
    res["loss"].backward()
    grad = rep.grad if rep.grad is not None else torch.zeros_like(rep)
    return cfg, inp, res, grad, (anchor_idx, neg_idx, highs)


@pytest.mark.parametrize("name", sorted(CASES_UNSUP))
def test_oracle_unsupervised_loss_matches_reference(name):
    z = np.load(GOLD)
    cfg = CASES_UNSUP[name]
    predict, target, teacher = unsup_inputs(cfg)
    predict.requires_grad_(True)
    loss, tgt = O.unsupervised_loss(predict, target, cfg["percent"], teacher)
    loss.backward()
    assert np.array_equal(tgt.numpy(), z[f"u_{name}_target"])
    assert abs(loss.item() - z[f"u_{name}_loss"][0]) <= 1e-6 * abs(z[f"u_{name}_loss"][0])
    assert np.allclose(predict.grad.numpy(), z[f"u_{name}_grad"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", sorted(CASES_CONTRA))
def test_oracle_contra_memobank_loss_matches_reference(name):
    z = np.load(GOLD)
    cfg, inp, res, grad, (_, _, highs) = run_oracle_contra(name)
    K = cfg["K"]
    assert np.array_equal(highs, z[f"c_{name}_draws"])      # same draws, from the same ranges, as the reference made
    assert res["new_keys"] == z[f"c_{name}_new_keys"].tolist()
    assert res["ptrs"] == z[f"c_{name}_ptrs"].tolist()
    assert [b.shape[0] for b in res["memobank"]] == z[f"c_{name}_bank_rows"].tolist()
    for c in range(K):
        assert abs(res["memobank"][c].double().sum().item() - z[f"c_{name}_bank_sum"][c]) <= 1e-9 + 1e-12 * abs(z[f"c_{name}_bank_sum"][c])
        if res["memobank"][c].shape[0]:
            assert np.array_equal(res["memobank"][c][-1, :4].numpy(), z[f"c_{name}_bank_last"][c])
    ref_loss = z[f"c_{name}_loss"][0]
    assert abs(res["loss"].item() - ref_loss) <= 1e-6 * max(abs(ref_loss), 1e-6), (res["loss"].item(), ref_loss)
    assert np.allclose(grad.numpy()[:, :16], z[f"c_{name}_grad"], rtol=2e-5, atol=1e-8)
    gn = np.sqrt((grad.double().numpy() ** 2).sum())
    assert abs(gn - z[f"c_{name}_gnorm"][0]) <= 1e-5 * max(z[f"c_{name}_gnorm"][0], 1e-9)
    if cfg.get("momentum"):
        assert abs(res["prototype"].double().sum().item() - z[f"c_{name}_prototype_sum"][0]) <= 1e-6 * abs(z[f"c_{name}_prototype_sum"][0])


def test_oracle_enqueue_is_a_sliding_window():
    q = torch.arange(12, dtype=torch.float32).view(6, 2)
    keys = torch.arange(100, 110, dtype=torch.float32).view(5, 2)
    out, ptr, m = O.memobank_enqueue(keys, q, 6, 8)
    assert m == 5 and ptr == 8 and out.shape[0] == 8
    assert torch.equal(out, torch.cat((q, keys))[-8:])
    out, ptr, m = O.memobank_enqueue(keys[:1], q, 6, 8)
    assert (ptr, out.shape[0]) == (7, 7)
