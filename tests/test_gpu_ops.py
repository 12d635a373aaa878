"""GPU parity, op by op, through the C ABI: BaseNet2 forward/backward, the loss block and Adam
against the CPU oracle on the same seeded inputs.  fp32; tolerances stated per check."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cmlpl_oracle as O
from tests.gpu_util import ModuleRegions, hip_relu_gates, relu_mask_audit, DEV, report, to_shape

pytestmark = pytest.mark.gpu

SHAPES = {
    "P": O.NetShape(60, 20, 20, 103, 9),        # reference-exact
    "B2": O.NetShape(103, 11, 11, 103, 9),      # BASELINE configs[1]
    "B4": O.NetShape(200, 11, 11, 200, 16),
    "B5": O.NetShape(48, 15, 15, 48, 20),
    # other window sizes: every column-pair count of the row-split weight-gradient kernel between 2 and 11,
    # even and odd maps, tiny and ragged channel counts
    "W8": O.NetShape(8, 8, 8, 12, 5),
    "W9": O.NetShape(33, 9, 9, 7, 4),
    "W12": O.NetShape(5, 12, 12, 9, 3),
    "W13": O.NetShape(5, 13, 13, 7, 4),
    "W16": O.NetShape(6, 16, 16, 10, 3),
    "W18": O.NetShape(3, 18, 18, 5, 3),
}


def _module(shape, params, dropout):
    from cmlpl_amd.models import BaseNet2
    net = BaseNet2(num_features=shape.bands, dropout=dropout, num_classes=shape.K,
                   in_channels=shape.C, window=shape.H).to(DEV)
    net.load_state_dict(params)
    return net


@pytest.mark.parametrize("name,n", [("P", 5), ("P", 64), ("B2", 37), ("B2", 256), ("B4", 16), ("B5", 23),
                                    ("W8", 19), ("W9", 50), ("W12", 11), ("W13", 7), ("W16", 9), ("W18", 6),
                                    ("B2", 1), ("B2", 300)])
def test_basenet2_forward_backward(name, n):
    shape = SHAPES[name]
    params = O.closed_form_params(shape, 7)
    g = torch.Generator().manual_seed(100 + n)
    x = torch.randn(n, shape.C, shape.H, shape.W, generator=g)
    y = torch.randn(n, shape.bands, generator=g)
    keep = 0.2
    dm = (torch.rand(n, shape.cls_in, generator=g) < keep).float() / keep
    dlog = torch.randn(n, shape.K, generator=g)
    dfe = torch.randn(n, 1024, generator=g) * 0.1
    # HIP forward
    net = _module(shape, params, dropout=0.8)
    net.train()
    lo, fe = net(x.to(DEV), y.to(DEV), dropmask=dm.to(DEV))
    torch.cuda.synchronize()
    # oracle, taking the device's ReLU decisions: among ~1e6 pre-activations a few sit within rounding of zero,
    # where the summation order decides the sign; a flipped gate is a different (equally valid) gradient.  The
    # audit requires every disagreement to sit at |z| < 2e-5.
    regions = ModuleRegions(net, lo, n)
    gates = hip_relu_gates(regions, shape, n)[0]
    pr = {k: v.clone().requires_grad_(k in O.LIVE_KEYS) for k, v in params.items()}
    taps = {}
    lo_ref, fe_ref = O.basenet2_forward(pr, x, y, dm, taps=taps, relu_gates=gates)
    (lo_ref * dlog).sum().add((fe_ref * dfe).sum()).backward()
    flips = relu_mask_audit(regions, [taps], shape, n)[0]
    assert sum(flips.values()) <= 4, flips
    report("logits", lo, lo_ref, 1e-4, 2e-5)
    report("feat", fe, fe_ref, 1e-5, 1e-6)
    ((lo * dlog.to(DEV)).sum() + (fe * dfe.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    hip = dict(net.named_parameters())
    for k in O.LIVE_KEYS:
        scale = float(pr[k].grad.abs().max())
        report("grad " + k, hip[k].grad, pr[k].grad, 2e-4, 2e-5 * max(scale, 1e-3))
    for k in ("feat_ss.weight", "feat_ss2.bias", "feat_ss3.weight"):
        assert hip[k].grad is None          # dead parameters, like the reference


@pytest.mark.parametrize("name,n", [("B2", 256), ("P", 64), ("W16", 9)])
def test_split_bf16_convolutions_keep_fp32_accuracy(name, n):
    """The 3x3 convolutions take fp32 operands as three bf16 pieces each and sum the six significant products on the
    bf16 MFMA (conv3x3.hip, "fp32 on the bf16 MFMA").  Measured against an fp64 evaluation of the oracle (with the
    device's ReLU decisions), every output and gradient must be as close as fp32 arithmetic gets: max error below
    2e-6 of the tensor's largest element (the f32-input MFMA kernels measured 1e-7 .. 6e-7 on the same inputs)."""
    shape = SHAPES[name]
    params = O.closed_form_params(shape, 7)
    g = torch.Generator().manual_seed(100 + n)
    x = torch.randn(n, shape.C, shape.H, shape.W, generator=g)
    y = torch.randn(n, shape.bands, generator=g)
    dm = (torch.rand(n, shape.cls_in, generator=g) < 0.2).float() / 0.2
    dlog = torch.randn(n, shape.K, generator=g)
    dfe = torch.randn(n, 1024, generator=g) * 0.1
    net = _module(shape, params, dropout=0.8)
    net.train()
    lo, fe = net(x.to(DEV), y.to(DEV), dropmask=dm.to(DEV))
    gates = hip_relu_gates(ModuleRegions(net, lo, n), shape, n)[0]
    ((lo * dlog.to(DEV)).sum() + (fe * dfe.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    pr = {k: v.clone().double().requires_grad_(k in O.LIVE_KEYS) for k, v in params.items()}
    lo_ref, fe_ref = O.basenet2_forward(pr, x.double(), y.double(), dm.double(), relu_gates=gates)
    (lo_ref * dlog.double()).sum().add((fe_ref * dfe.double()).sum()).backward()

    def rel(a, b):
        a = a.detach().cpu().double()
        return float((a - b.detach()).abs().max() / b.detach().abs().max())
    errs = {"logits": rel(lo, lo_ref), "feat": rel(fe, fe_ref)}
    hip = dict(net.named_parameters())
    for k in O.LIVE_KEYS:
        errs["grad " + k] = rel(hip[k].grad, pr[k].grad)
    print({k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < 2e-6, errs


@pytest.mark.parametrize("mixed", [False, True], ids=["unit-scale", "mixed-1e-6..1e3"])
def test_split_bf16_conv1_error_bound_per_element(mixed):
    """Per-ELEMENT accuracy of the split-bf16 3x3 convolution (forward, and the weight gradient), not just against the
    tensor's maximum.  conv1's output and weight gradient are re-computed in fp64 FROM THE DEVICE'S OWN operands (the
    saved a0, the saved pooled gradient and ReLU masks), so the only difference is conv1's arithmetic.  The bound is
    the one an fp32 dot product obeys, |err_i| <= c * S_i with S_i = sum_k |a_k| |w_k| (+ |bias| + |residual|) the
    absolute-value convolution of that element -- a bound relative to the element's own terms, with no floor borrowed
    from larger elements.  A split product drops at most 2^-21 |a w| and the fp32 accumulation
    adds its rounding; stated bounds: forward (K = 576 terms) c = 2^-18, weight gradient (K = n * 121 terms) c = 2^-16;
    the measured maxima are printed.  `mixed`: the 64 channels of a0 span 1e-6 .. 1e3 (conv0 scaled per channel, conv1's
    input channels scaled inversely): operands of very different magnitude meet in every sum, outputs of different
    channels differ by nine orders of magnitude, and every element still has to meet its own bound."""
    import torch.nn.functional as F
    shape, n = SHAPES["B2"], 32
    params = O.closed_form_params(shape, 9)
    if mixed:
        g0 = torch.Generator().manual_seed(5)
        sc = 10.0 ** (torch.rand(64, generator=g0) * 9.0 - 6.0)
        params["conv0.weight"] = params["conv0.weight"] * sc.view(64, 1, 1, 1)
        params["conv0.bias"] = params["conv0.bias"] * sc
        params["conv1.weight"] = params["conv1.weight"] / sc.view(1, 64, 1, 1)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, shape.C, shape.H, shape.W, generator=g)
    y = torch.randn(n, shape.bands, generator=g)
    dlog = torch.randn(n, shape.K, generator=g)
    net = _module(shape, params, dropout=0.0)
    net.train()
    lo, fe = net(x.to(DEV), y.to(DEV))
    reg = ModuleRegions(net, lo, n)
    (lo * dlog.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    H, W, H2, W2 = shape.H, shape.W, shape.H // 2, shape.W // 2
    a0 = reg.debug_region("a0").view(1, n, H * W, 64)[0].cpu().double().permute(0, 2, 1).reshape(n, 64, H, W)
    p1 = reg.debug_region("p1").view(1, n, H2 * W2, 64)[0].cpu().double().permute(0, 2, 1).reshape(n, 64, H2, W2)
    dp1 = reg.debug_region("dp1").view(1, n, H2 * W2, 64)[0].cpu().double().permute(0, 2, 1).reshape(n, 64, H2, W2)
    gate = hip_relu_gates(reg, shape, n)[0]["z1"]                                   # [n, 64, H, W] bool
    w1, b1 = params["conv1.weight"].double(), params["conv1.bias"].double()
    # ---- forward: p1 = avgpool2(relu(conv1(a0) + b1 + a0)), the device's ReLU decisions
    z1 = F.conv2d(a0, w1, b1, padding=1) + a0
    s1 = F.conv2d(a0.abs(), w1.abs(), b1.abs(), padding=1) + a0.abs()
    zero = torch.zeros((), dtype=torch.float64)
    p1_ref = F.avg_pool2d(torch.where(gate, z1, zero), 2)
    s_p1 = F.avg_pool2d(torch.where(gate, s1, zero), 2)
    err = (p1 - p1_ref).abs()
    ratio_f = float((err / s_p1.clamp_min(1e-300)).max())
    # ---- weight gradient: dW1 = sum a0(shifted) * dz1, dz1 = mask * upsample(dp1) / 4 (avgpool + ReLU backward)
    dz1 = torch.zeros(n, 64, H, W, dtype=torch.float64)
    dz1[:, :, :2 * H2, :2 * W2] = dp1.repeat_interleave(2, 2).repeat_interleave(2, 3) / 4
    dz1 = torch.where(gate, dz1, zero)
    dz1[:, :, 2 * H2:, :] = 0
    dz1[:, :, :, 2 * W2:] = 0
    dw_ref = torch.nn.grad.conv2d_weight(a0, w1.shape, dz1, padding=1)
    s_dw = torch.nn.grad.conv2d_weight(a0.abs(), w1.shape, dz1.abs(), padding=1)
    dw = dict(net.named_parameters())["conv1.weight"].grad.cpu().double()
    errw = (dw - dw_ref).abs()
    ratio_w = float((errw / s_dw.clamp_min(1e-300)).max())
    print(f"[{'mixed' if mixed else 'unit'}] forward: max err/S = {ratio_f:.3e} (2^{np.log2(max(ratio_f, 1e-300)):.1f}), "
          f"|p1| from {float(p1_ref.abs()[p1_ref != 0].min()):.2e} to {float(p1_ref.abs().max()):.2e};  "
          f"weight gradient: max err/S = {ratio_w:.3e} (2^{np.log2(max(ratio_w, 1e-300)):.1f})")
    assert ratio_f <= 2.0 ** -18, ratio_f
    assert ratio_w <= 2.0 ** -16, ratio_w


def test_window_too_large_for_lds_fails_loudly():
    """a 22x22 window needs a 24x24x68-float LDS image (157 KB) + tap buffer: no silent fallback, a shape error"""
    from cmlpl_amd import _lib
    shape = O.NetShape(4, 22, 22, 6, 3)
    net = _module(shape, O.closed_form_params(shape, 1), dropout=0.0)
    with pytest.raises(_lib.CmlplError):
        net(torch.zeros(2, 4, 22, 22, device=DEV), torch.zeros(2, 6, device=DEV))


def test_basenet2_eval_matches_oracle_and_state_dict_keys():
    shape = SHAPES["P"]
    params = O.closed_form_params(shape, 3)
    net = _module(shape, params, dropout=0.8)
    assert list(net.state_dict().keys()) == list(O.param_shapes(shape).keys())
    for k, shp in O.param_shapes(shape).items():
        assert tuple(net.state_dict()[k].shape) == tuple(shp)
    net.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(33, 60, 20, 20, generator=g)
    y = torch.randn(33, 103, generator=g)
    with torch.no_grad():
        lo, fe = net(x.to(DEV), y.to(DEV))
    lo_ref, fe_ref = O.basenet2_forward(params, x, y, None)
    report("logits", lo, lo_ref, 1e-4, 2e-5)
    report("feat", fe, fe_ref, 1e-5, 1e-6)
    assert torch.equal(lo.argmax(1).cpu(), lo_ref.argmax(1))


@pytest.mark.parametrize("name", ["B2", "P"])
def test_non_finite_input_stays_in_its_own_sample(name):
    """An Inf / NaN pixel: the three-piece bf16 split turns +-Inf into NaN (Inf - Inf in the remainder), the
    reference would carry Inf into sums of mixed-sign products, i.e. NaN one layer later.  Either way the damaged sample's
    logits are not finite in the oracle AND on the device, and every other sample of the batch is untouched (the
    per-sample kernels share nothing across samples; the general kernels' tiles may hold several samples)."""
    shape = SHAPES[name]
    params = O.closed_form_params(shape, 3)
    net = _module(shape, params, dropout=0.8)
    net.eval()
    g = torch.Generator().manual_seed(11)
    n = 19
    x = torch.randn(n, shape.C, shape.H, shape.W, generator=g)
    y = torch.randn(n, shape.bands, generator=g)
    x[3, 5, 2, 2] = float("inf")
    x[11, 0, 0, 0] = float("nan")
    x[17, shape.C - 1, shape.H - 1, shape.W - 1] = float("-inf")
    with torch.no_grad():
        lo, fe = net(x.to(DEV), y.to(DEV))
    lo_ref, fe_ref = O.basenet2_forward(params, x, y, None)
    bad = [3, 11, 17]
    good = [i for i in range(n) if i not in bad]
    for i in bad:
        assert not torch.isfinite(lo_ref[i]).all(), i                     # the reference arithmetic does not survive it either
        assert not torch.isfinite(lo[i].cpu()).all(), i
    report("logits of the undamaged samples", lo[good], lo_ref[good], 1e-4, 2e-5)
    report("features of the undamaged samples", fe[good], fe_ref[good], 1e-5, 1e-6)


def test_dropout_philox_statistics():
    shape = SHAPES["B2"]
    net = _module(shape, O.closed_form_params(shape, 3), dropout=0.8)
    net.train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(64, 103, 11, 11, generator=g).to(DEV)
    y = torch.randn(64, 103, generator=g).to(DEV)
    lo1, _ = net(x, y)
    lo2, _ = net(x, y)
    net.eval()
    lo3, _ = net(x, y)
    assert torch.isfinite(lo1).all()
    assert not torch.allclose(lo1, lo2)          # fresh mask per call
    assert not torch.allclose(lo1, lo3)


def _loss_inputs(shape, bt, btu, Q, seed, peaky):
    g = torch.Generator().manual_seed(seed)
    n = bt + btu
    sc = 6.0 if peaky else 1.0
    z = [torch.randn(n, shape.K, generator=g) * sc for _ in range(2)]
    if peaky:   # make the two nets agree on many rows so Q0 crosses 0.8
        z[1] = z[0] + 0.3 * torch.randn(n, shape.K, generator=g)
    f = [O.l2norm(torch.relu(torch.randn(n, 1024, generator=g))) for _ in range(2)]
    Y = torch.randint(0, shape.K, (bt,), generator=g)
    bf = [O.l2norm(torch.relu(torch.randn(Q, 1024, generator=g))) for _ in range(2)]
    bp = [torch.softmax(torch.randn(Q, shape.K, generator=g) * 3, 1) for _ in range(2)]
    return z, f, Y, bf, bp


@pytest.mark.parametrize("bt,btu,smooth,peaky,K", [(32, 32, True, False, 9), (128, 128, True, True, 9),
                                                   (16, 48, False, True, 9), (24, 40, True, True, 20),
                                                   (128, 128, False, False, 16), (20, 44, True, False, 9),
                                                   (10, 42, True, True, 9), (8, 120, True, False, 12)])
def test_loss_block(bt, btu, smooth, peaky, K):
    from cmlpl_amd import _lib
    lib = _lib.load()
    shape = O.NetShape(60, 20, 20, 103, K)
    hp = O.HyperParams()
    Q = 10 * bt if 10 * bt >= bt + btu else 4 * (bt + btu)
    n = bt + btu
    z, f, Y, bf, bp = _loss_inputs(shape, bt, btu, Q, 1000 + bt + btu, peaky)
    adap = 0.9 if peaky else 1.0
    zr = [t.clone().requires_grad_(True) for t in z]
    fr = [t.clone().requires_grad_(True) for t in f]
    lb = O.loss_block(zr[0], fr[0], zr[1], fr[1], Y, bt, bf, bp, smooth, adap, hp)
    gs = torch.autograd.grad(lb["total_s"], [zr[0], fr[0]], allow_unused=True)
    gw = torch.autograd.grad(lb["total_w"], [zr[1], fr[1]], allow_unused=True)
    # HIP
    cs = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
    d = lambda t: t.to(DEV).contiguous()
    logits, feat, labels = d(torch.stack(z)), d(torch.stack(f)), d(Y)
    bank_f, bank_p = [d(t.clone()) for t in bf], [d(t.clone()) for t in bp]
    banks = _lib.Banks()
    ptr = [Q - 5, 3]          # bank0 write wraps around the end
    for i in range(2):
        banks.d_feats[i] = bank_f[i].data_ptr(); banks.d_probs[i] = bank_p[i].data_ptr(); banks.ptr[i] = ptr[i]
    banks.Q = Q
    chp = _lib.HParams(hp.lr, hp.beta1, hp.beta2, hp.eps, hp.temperature, hp.alpha, hp.noise, hp.dropout,
                       hp.w_contrast, hp.w_mutual, hp.pos_thr, hp.neg_thr)
    scal = torch.zeros(16, device=DEV)
    dlog, dfe = torch.full((2, n, K), 7.0, device=DEV), torch.full((2, n, 1024), 7.0, device=DEV)
    probs = torch.zeros(4, btu, K, device=DEV)
    wsb = lib.cmlpl_workspace_bytes(C.byref(cs), 2, n, Q)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    rc = lib.cmlpl_loss_fwd_bwd(C.byref(cs), bt, btu, logits.data_ptr(), feat.data_ptr(), labels.data_ptr(),
                                C.byref(banks), int(smooth), adap, C.byref(chp), scal.data_ptr(), dlog.data_ptr(),
                                dfe.data_ptr(), probs.data_ptr(), ws.data_ptr(), ws.numel(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    s = scal.cpu().numpy()
    want = [lb[k].item() for k in ("ctr_s", "total_s", "cls_s", "con_s", "acc", "total_w", "cls_w", "con_w", "ctr_w")]
    report("scalars", s[:9], np.array(want), 1e-4, 1e-6)        # loss parity: 1e-4 relative (north_star)
    cnt = [lb["mask_w"].sum().item(), lb["mask_s"].sum().item(), (lb["Q"] > 0).sum().item()]
    assert list(s[9:12]) == cnt, (s[9:13], cnt)
    report("probs", probs, torch.stack([lb["p_w"], lb["p_s"], lb["p_w0"], lb["p_s0"]]), 1e-4, 1e-6)
    report("dlogits_s", dlog[0], gs[0], 2e-4, 1e-7)
    report("dlogits_w", dlog[1], gw[0], 2e-4, 1e-7)
    zero = torch.zeros(n, 1024)
    report("dfeat_s", dfe[0], gs[1] if gs[1] is not None else zero, 5e-4, 2e-7)
    report("dfeat_w", dfe[1], gw[1] if gw[1] is not None else zero, 5e-4, 2e-7)
    # bank write (train.py:232-236), modulo Q
    for i, rows in enumerate((lb["bank0_rows"], lb["bank1_rows"])):
        ef, ep = bf[i].clone(), bp[i].clone()
        O.bank_write(ef, ptr[i], rows[0]); O.bank_write(ep, ptr[i], rows[1])
        report(f"bank{i}_feats", bank_f[i], ef, 0, 0)            # pure copy: bit-exact
        report(f"bank{i}_probs", bank_p[i], ep, 1e-5, 1e-7)


def test_adam_step_matches_torch_formula():
    from cmlpl_amd import _lib
    lib = _lib.load()
    shape = SHAPES["B2"]
    cs = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
    L = _lib.layout(cs)
    P, live = int(L.param_total), int(L.param_live)
    hp = O.HyperParams()
    chp = _lib.HParams(hp.lr, hp.beta1, hp.beta2, hp.eps, hp.temperature, hp.alpha, hp.noise, hp.dropout,
                       hp.w_contrast, hp.w_mutual, hp.pos_thr, hp.neg_thr)
    g = torch.Generator().manual_seed(4)
    p = torch.randn(2, P, generator=g) * 0.05
    pc, m, v = p.clone(), torch.zeros(2, P), torch.zeros(2, P)
    dp, dm, dv = p.to(DEV), torch.zeros(2, P, device=DEV), torch.zeros(2, P, device=DEV)
    for t in range(1, 5):
        gr = torch.randn(2, P, generator=g) * (10.0 ** (t - 3))
        rc = lib.cmlpl_adam_step(C.byref(cs), 2, dp.data_ptr(), P, gr.to(DEV).data_ptr(), P, dm.data_ptr(),
                                 dv.data_ptr(), t, C.byref(chp), None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        for net in range(2):
            O.adam_update(pc[net, :live], gr[net, :live], m[net, :live], v[net, :live], t, hp)
        torch.cuda.synchronize()
        report(f"adam t={t} params", dp[:, :live], pc[:, :live], 1e-6, 1e-7)
        assert torch.equal(dp[:, live:].cpu(), p[:, live:])          # dead tensors untouched


@pytest.mark.parametrize("name", ["B2", "P", "B4"])
def test_adam_refreshes_every_packed_weight_copy_like_pack_weights(name):
    """The optimizer rewrites the kernel-side weight copies itself (split-bf16 fragment sets of the 3x3 and conv0
    weights, conv2's f32 fragments, the k-major thin weights): after a step the packed buffer must equal, bit for
    bit, what cmlpl_pack_weights makes of the updated parameters."""
    from cmlpl_amd import _lib
    lib = _lib.load()
    shape = SHAPES[name]
    cs = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
    L = _lib.layout(cs)
    P, PK = int(L.param_total), int(L.packed_total)
    hp = O.HyperParams()
    chp = _lib.HParams(hp.lr, hp.beta1, hp.beta2, hp.eps, hp.temperature, hp.alpha, hp.noise, hp.dropout,
                       hp.w_contrast, hp.w_mutual, hp.pos_thr, hp.neg_thr)
    g = torch.Generator().manual_seed(11)
    dp = (torch.randn(2, P, generator=g) * 0.05).to(DEV)
    dm, dv = torch.zeros(2, P, device=DEV), torch.zeros(2, P, device=DEV)
    packed = torch.full((2, PK), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    # as the engines do: one cmlpl_pack_weights when parameters are loaded (it also zeroes the pad rows, which the
    # optimizer never touches), then the optimizer keeps the copies current
    assert lib.cmlpl_pack_weights(C.byref(cs), 2, dp.data_ptr(), P, packed.data_ptr(), st) == 0
    for t in range(1, 3):
        gr = (torch.randn(2, P, generator=g) * 0.01).to(DEV)
        assert lib.cmlpl_adam_step(C.byref(cs), 2, dp.data_ptr(), P, gr.data_ptr(), P, dm.data_ptr(), dv.data_ptr(), t,
                                   C.byref(chp), packed.data_ptr(), st) == 0
    want = torch.empty(2, PK, device=DEV)
    assert lib.cmlpl_pack_weights(C.byref(cs), 2, dp.data_ptr(), P, want.data_ptr(), st) == 0
    torch.cuda.synchronize()
    a, b = packed.view(torch.int32).cpu(), want.view(torch.int32).cpu()
    diff = (a != b)
    assert not diff.any(), f"{int(diff.sum())} packed words differ, first at {diff.nonzero()[0].tolist()}"


def test_whole_image_inference_matches_oracle():
    """tools.hyper_tools.test_whole (reference hyper_tools.py:416-437) over a 'wholeset'
    loader with a ragged last batch: same argmax as the oracle's eval forward for every pixel."""
    from torch.utils.data import DataLoader
    from hsi_loader import SyntheticHSIDataSet
    from tools.hyper_tools import CalAccuracy, test_whole
    shape = SHAPES["B2"]
    params = O.closed_form_params(shape, 9)
    net = _module(shape, params, dropout=0.8)
    ds = SyntheticHSIDataSet((shape.C, shape.H, shape.W, shape.bands, shape.K), 1100, "wholeset", seed=3)
    pred = test_whole(net, DataLoader(ds, batch_size=512, shuffle=False), print_per_batches=10 ** 9)
    lo_ref, _ = O.basenet2_forward(params, ds.XP, ds.X, None)
    ref = lo_ref.argmax(1).numpy()
    assert pred.shape == (1100,)
    top2 = lo_ref.topk(2, 1)[0]
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-4).numpy()            # ignore fp32 near-ties
    assert np.array_equal(pred[clear], ref[clear]) and clear.mean() > 0.99
    OA, Kappa, pa = CalAccuracy(pred, ref)
    assert OA > 0.99 and pa.shape == (shape.K,)
