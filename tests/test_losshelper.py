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
    rows = z[f"u_{name}_grad"].shape[0]                                  # big cases keep the first rows + the norm
    assert np.allclose(predict.grad.numpy()[:rows], z[f"u_{name}_grad"], rtol=1e-5, atol=1e-8)
    gn = np.sqrt((predict.grad.double().numpy() ** 2).sum())
    assert abs(gn - z[f"u_{name}_gnorm"][0]) <= 1e-6 * z[f"u_{name}_gnorm"][0]


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


# ------------------------------------------------------------------------------------------ GPU parity
def _bank_from_inputs(inp, cfg, dev):
    from cmlpl_amd.memobank import MemoryBank
    ptrs = [torch.tensor([p], dtype=torch.long) for p in inp["ptrs"]]
    return MemoryBank.from_lists([[b] for b in inp["bank"]], ptrs, inp["sizes"], cfg["D"], dev)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES_UNSUP))
def test_hip_unsupervised_loss_matches_oracle(name):
    import loss_helper as LH
    cfg = CASES_UNSUP[name]
    predict, target, teacher = unsup_inputs(cfg)
    pr = predict.clone().requires_grad_(True)
    ref_loss, ref_tgt = O.unsupervised_loss(pr, target.clone(), cfg["percent"], teacher)
    ref_loss.backward()
    gp = predict.cuda().requires_grad_(True)
    gt = target.clone().cuda()
    loss = LH.compute_unsupervised_loss(gp, gt, cfg["percent"], teacher.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(gt.cpu(), ref_tgt)                                   # the dropped set is exact
    assert abs(loss.item() - ref_loss.item()) <= 2e-6 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    assert torch.allclose(gp.grad.cpu(), pr.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES_CONTRA))
@pytest.mark.parametrize("style", ["bank", "lists"])
def test_hip_contra_memobank_loss_matches_oracle(name, style):
    import loss_helper as LH
    cfg, inp, res, grad, (anchor_idx, neg_idx, _) = run_oracle_contra(name)
    K, D = cfg["K"], cfg["D"]
    dev = torch.device("cuda")
    rep = inp["rep"].cuda().requires_grad_(True)
    args = [inp[k].cuda() for k in ("label_l", "label_u", "prob_l", "prob_u", "low_mask", "high_mask")]
    mp = inp["momentum"].cuda() if cfg.get("momentum") else None
    if style == "bank":
        bank = _bank_from_inputs(inp, cfg, dev)
        out = LH.compute_contra_memobank_loss(rep, *args, bank, None, None, inp["rep_teacher"].cuda(),
                                              momentum_prototype=mp, i_iter=cfg.get("i_iter", 0),
                                              _draws=(anchor_idx, neg_idx))
        rows_after = [bank.rows(c).cpu() for c in range(K)]
        ptrs_after = list(bank.ptrs)
    else:
        memobank = [[b.clone().cuda()] for b in inp["bank"]]
        ptrs = [torch.tensor([p], dtype=torch.long) for p in inp["ptrs"]]
        out = LH.compute_contra_memobank_loss(rep, *args, memobank, ptrs, list(inp["sizes"]),
                                              inp["rep_teacher"].cuda(), momentum_prototype=mp,
                                              i_iter=cfg.get("i_iter", 0), _draws=(anchor_idx, neg_idx))
        rows_after = [memobank[c][0].cpu() for c in range(K)]
        ptrs_after = [int(p[0]) for p in ptrs]
    if mp is None:
        new_keys, loss = out
    else:
        prototype, new_keys, loss = out
        assert torch.allclose(prototype.cpu(), res["prototype"], rtol=1e-5, atol=1e-6, equal_nan=True)
    loss.backward()
    torch.cuda.synchronize()
    assert new_keys == res["new_keys"]
    assert ptrs_after == res["ptrs"]
    for c in range(K):
        assert torch.equal(rows_after[c], res["memobank"][c]), f"bank of class {c}"     # pure copies: bit-exact
    ref = res["loss"].item()
    assert abs(loss.item() - ref) <= 1e-5 * max(abs(ref), 1e-6), (loss.item(), ref)
    g = rep.grad.cpu() if rep.grad is not None else torch.zeros_like(grad)
    err = (g - grad).abs().max().item()
    assert err <= 2e-4 * max(grad.abs().max().item(), 1e-9) + 1e-9, (err, grad.abs().max().item())


@pytest.mark.gpu
def test_in_kernel_draws_advance_from_call_to_call():
    """The drop-in list path builds a fresh MemoryBank on every call (loss_helper.compute_contra_memobank_loss with
    the reference's list structure): the in-kernel anchor / negative draws must still differ from one training
    iteration to the next (they are keyed by a 64-bit value drawn from torch's CPU generator per call), and a run
    must stay reproducible under torch.manual_seed."""
    import loss_helper as LH
    cfg, inp, res, grad, _ = run_oracle_contra("k9")
    args = [inp[k].cuda() for k in ("label_l", "label_u", "prob_l", "prob_u", "low_mask", "high_mask")]

    def one_call():
        rep = inp["rep"].cuda().requires_grad_(True)
        memobank = [[b.clone().cuda()] for b in inp["bank"]]
        ptrs = [torch.tensor([p], dtype=torch.long) for p in inp["ptrs"]]
        out = LH.compute_contra_memobank_loss(rep, *args, memobank, ptrs, list(inp["sizes"]), inp["rep_teacher"].cuda())
        out[-1].backward()
        torch.cuda.synchronize()
        return rep.grad.clone()

    torch.manual_seed(123)
    g1, g2 = one_call(), one_call()
    assert not torch.equal(g1, g2)                  # consecutive iterations draw different anchors / negatives
    torch.manual_seed(123)
    h1, h2 = one_call(), one_call()
    assert torch.equal(g1, h1) and torch.equal(g2, h2)      # same seed, same sequence of calls: same draws


@pytest.mark.gpu
def test_hip_enqueue_sliding_window():
    import loss_helper as LH
    rng = np.random.Generator(np.random.PCG64(9))
    q_ref = torch.zeros(0, 8)
    queue, ptr = [torch.zeros(0, 8).cuda()], torch.zeros(1, dtype=torch.long)
    p_ref = 0
    for m in (3, 0, 5, 7, 40, 1, 13):                      # capacity 12: fills, wraps, and one batch larger than it
        keys = torch.from_numpy(rng.standard_normal((m, 8)).astype(np.float32))
        q_ref, p_ref, n_ref = O.memobank_enqueue(keys, q_ref, p_ref, 12)
        n = LH.dequeue_and_enqueue(keys.cuda(), queue, ptr, 12)
        assert n == n_ref and int(ptr[0]) == p_ref
        assert torch.equal(queue[0].cpu(), q_ref)
