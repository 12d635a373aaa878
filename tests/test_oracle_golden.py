"""CPU: the oracle restatement vs fixtures produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import cmlpl_oracle as O
from tests.golden_util import GoldenCase, golden_cases, rel_err

RTOL = 2e-5      # same torch build, same op order: observed ~1e-7; slack for thread-count effects


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_reference_fixture(name):
    torch.set_num_threads(8)
    g = GoldenCase(name)
    p0, p1 = g.params()
    st = O.StepState.create(g.shape, p0, p1, g.bt, g.hp)
    for s in range(g.steps):
        b = g.batch(s)
        epoch, bi = g.epoch_bi(s)
        out = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"],
                           b["dropmask"], epoch, bi, g.hp)
        z = g.z
        assert rel_err(out["hist"], z["hist"][s], 1e-9) < RTOL, (s, out["hist"], z["hist"][s])
        extra = [float(out["total_w"]), float(out["cls_w"]), float(out["con_w"]), float(out["ctr_w"])]
        assert rel_err(extra, z["extra"][s], 1e-9) < RTOL, (s, extra, z["extra"][s])
        assert list(st.ptr) == [int(v) for v in z["ptr"][s]], (s, st.ptr, z["ptr"][s])
        counts = [float(out["mask_w"].sum()), float(out["mask_s"].sum()), out["n_pos"], out["n_neg"]]
        assert counts == list(z["counts"][s]), (s, counts, z["counts"][s])
        ls = [float(out["logits"][0].sum()), float(out["logits"][0].abs().sum()),
              float(out["logits"][1].sum()), float(out["logits"][1].abs().sum())]
        assert rel_err(ls[1::2], z["logit_sums"][s][1::2]) < RTOL
        for net in range(2):
            gn = [float(out["grads"][net][k].double().norm()) for k in O.LIVE_KEYS]
            assert rel_err(gn, z["grad_norms"][s][net], 1e-9) < 5e-5, (s, net, gn, z["grad_norms"][s][net])
            psum = [float(st.params[net][k].double().sum()) for k in O.LIVE_KEYS]
            assert np.allclose(psum, z["param_sums"][s][net], rtol=1e-5, atol=1e-5, equal_nan=True), (s, net)
            if "grad_nan" in z.files:     # NaN lands on the same number of elements of every gradient tensor
                assert [int(torch.isnan(out["grads"][net][k]).sum()) for k in O.LIVE_KEYS] == list(z["grad_nan"][s][net])
        bs = [float(st.bank_feats[0].double().sum()), float(st.bank_probs[0].double().sum()),
              float(st.bank_feats[1].double().sum()), float(st.bank_probs[1].double().sum())]
        assert np.allclose(bs, z["bank_sums"][s], rtol=1e-5, atol=1e-4, equal_nan=True), (s, bs, z["bank_sums"][s])
        if "bank_nan" in z.files:
            assert [int(torch.isnan(t).sum()) for t in (st.bank_feats[0], st.bank_probs[0], st.bank_feats[1],
                                                        st.bank_probs[1])] == list(z["bank_nan"][s])
        if s in g.full_steps:
            assert np.allclose(torch.stack(out["logits"]).numpy(), z[f"s{s}_logits"], rtol=1e-4, atol=2e-5, equal_nan=True)
            f8 = np.stack([out["feats"][0].numpy()[:, :8], out["feats"][1].numpy()[:, :8]])
            assert np.allclose(f8, z[f"s{s}_feats8"], rtol=1e-5, atol=1e-6, equal_nan=True)
            pr = np.stack([out["p_w"].numpy(), out["p_s"].numpy()])
            assert np.allclose(pr, z[f"s{s}_probs"], rtol=1e-4, atol=1e-6, equal_nan=True)
            mk = np.stack([out["mask_w"].numpy(), out["mask_s"].numpy()])
            assert np.array_equal(mk, z[f"s{s}_masks"])
            qd = np.stack([out["Q"].diag().numpy(), out["Qn"].sum(1).numpy()])
            assert np.allclose(qd, z[f"s{s}_Qdiag"], rtol=1e-4, atol=1e-6, equal_nan=True)
            gc = np.stack([out["grads"][0]["classifier.weight"].numpy()[:, :16],
                           out["grads"][1]["classifier.weight"].numpy()[:, :16]])
            assert np.allclose(gc, z[f"s{s}_grad_cls"], rtol=1e-4, atol=1e-6, equal_nan=True)


def test_adam_restatement_matches_torch_optim():
    torch.manual_seed(0)
    hp = O.HyperParams()
    p = torch.randn(257)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=hp.lr)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for t in range(1, 6):
        g = torch.randn(257)
        ref.grad = g.clone()
        opt.step()
        O.adam_update(p, g, m, v, t, hp)
        assert torch.allclose(p, ref.detach(), rtol=0, atol=1e-7)


def test_bank_pointer_quirk_and_wrap():
    # train.py:234,237 -- ptr1 follows ptr0, both advance by the literal 256
    q = 1280
    ptr = [0, 0]
    seen = []
    for _ in range(6):
        p0 = (ptr[0] + 256) % q
        ptr = [p0, (p0 + 256) % q]
        seen.append(tuple(ptr))
    assert seen[-1] == (256, 512)
    bank = torch.zeros(10, 2)
    O.bank_write(bank, 8, torch.ones(4, 2))
    assert bank[:, 0].tolist() == [1, 1, 0, 0, 0, 0, 0, 0, 1, 1]
