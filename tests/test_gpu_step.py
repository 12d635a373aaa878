"""GPU parity of the whole training step (cmlpl_train_step) against
  (a) the committed golden fixtures produced by the reference itself, and
  (b) the CPU oracle run side by side on the same seeded inputs (explicit noise / dropout masks),
plus size-independent properties at BASELINE.json's full batch."""
import numpy as np
import pytest
import torch

from oracle import cmlpl_oracle as O
from tests.golden_util import GoldenCase, golden_cases, rel_err
from tests.gpu_util import (DEV, cuda_batch, hip_relu_gates, relu_mask_audit, report, report_after_updates,
                            report_params, to_hp, to_shape)

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4          # north_star: loss parity within 1e-4 relative
GOLDEN_GRAD_RTOL = 2e-3   # gradient norms vs the reference run (other host / thread count): same ReLU decisions assumed


def _engine(g):
    from cmlpl_amd import TrainEngine
    eng = TrainEngine(to_shape(g.shape), g.bt, g.btu, to_hp(g.hp), device=DEV)
    p0, p1 = g.params()
    eng.load_state_dict(0, p0)
    eng.load_state_dict(1, p1)
    return eng, p0, p1


NAN_CASES = ("b2_deadrelu_64",)


@pytest.mark.parametrize("name", [n for n in golden_cases() if n not in NAN_CASES])
def test_step_matches_golden_and_oracle(name):
    g = GoldenCase(name)
    eng, p0, p1 = _engine(g)
    st = O.StepState.create(g.shape, p0, p1, g.bt, g.hp)
    z = g.z
    steps = min(g.steps, 8) if name != "p_traj_32" else g.steps
    total_flips = 0
    n = g.bt + g.btu
    for s in range(steps):
        b = g.batch(s)
        epoch, bi = g.epoch_bi(s)
        cb = cuda_batch(b)
        eng.step(cb["XPl"], cb["Xl"], cb["Y"], cb["XPu"], cb["Xu"], epoch, bi, noise=cb["noise"],
                 dropmask=cb["dropmask"])
        # The oracle takes the device's ReLU decisions (they can differ from sign(z_oracle) only where
        # |z| < 2e-5 -- audited below -- because fp32 summation order differs between MFMA and the CPU), so
        # every gradient is compared at the tight bound and both sides stay on one trajectory.
        gates = hip_relu_gates(eng, g.shape, n)
        ref = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"],
                           epoch, bi, g.hp, relu_gates=gates)
        sc = eng.read_scalars()
        row = [sc[k] for k in ("ctr_s", "total_s", "cls_s", "con_s", "acc")]
        extra = [sc[k] for k in ("total_w", "cls_w", "con_w", "ctr_w")]
        print(f"[{name}] step {s}: hip={row} golden={list(z['hist'][s])}")
        # (a) golden fixture (reference outputs)
        assert rel_err(row, z["hist"][s], 1e-7) < LOSS_RTOL, (s, row, z["hist"][s])
        assert rel_err(extra, z["extra"][s], 1e-7) < LOSS_RTOL, (s, extra, z["extra"][s])
        assert eng.ptr == [int(v) for v in z["ptr"][s]]
        assert [sc["n_mask_w"], sc["n_mask_s"], sc["n_pos"], sc["n_neg"]] == list(z["counts"][s])
        # (b) oracle, tensor by tensor
        lo, fe = eng.outputs()
        lo_ref = torch.stack(ref["logits"])
        report("logits", lo, lo_ref, 2e-4, 5e-6 * float(lo_ref.abs().max()) + 2e-5)
        report_after_updates("feat", fe, torch.stack(ref["feats"]), 1e-5, 3e-6, s, g.hp.lr)
        # ReLU masks saved by the HIP forward vs the oracle's pre-activation signs: a mismatch is only
        # tolerated exactly at the activation boundary (|z| < 2e-5), where fp32 summation order decides -- at step 0.
        # After s Adam updates a weight whose gradient is ~eps-sized may sit O(lr) away from the oracle's (Adam moves it
        # by ~lr whatever the gradient's size; its SIGN is decided by rounding: see report_params), which shifts the
        # pre-activations it feeds by O(lr * |a|): the boundary widens by 0.1 lr per step for the convolutions' masks
        # as it does by 0.25 lr for the spectral branch's
        flips = relu_mask_audit(eng, ref["taps"], g.shape, n, ztol=2e-5 + 0.1 * g.hp.lr * s, ztol_y=2e-5 + 0.25 * g.hp.lr * s)
        total_flips += sum(sum(f.values()) for f in flips)
        for net in range(2):
            for k in O.LIVE_KEYS:
                gr = ref["grads"][net][k]
                mx = max(float(gr.abs().max()), 1e-4)
                report(f"grad[{net}] {k}", eng.grad(net, k), gr, 5e-4, 5e-5 * mx)
            if not any(flips[net].values()):
                # the golden run took sign(z) decisions: comparable whenever the device took the same ones
                gn = [float(eng.grad(net, k).double().norm()) for k in O.LIVE_KEYS]
                # the two trajectories (this device, the host that produced the fixture) drift apart slowly: the bound
                # grows with the step index (20-step case: 2e-3 at step 0, 7e-3 at step 19)
                assert rel_err(gn, z["grad_norms"][s][net], 1e-9) < GOLDEN_GRAD_RTOL * (1 + s / 8), (s, net, gn, z["grad_norms"][s][net])
        if s in g.full_steps:
            # later steps of the golden run carry another host's ReLU-boundary decisions (see above)
            gtol = 2e-5 if s == 0 else 5e-3
            report("golden logits", lo, z[f"s{s}_logits"], 2e-4, 5e-6 * float(lo_ref.abs().max()) + gtol)
    assert total_flips <= 2 * steps, total_flips          # flips are rare events, not the norm
    # parameters and banks after the trajectory
    for net in range(2):
        sd = eng.state_dict(net)
        for k in O.LIVE_KEYS:
            report_params(f"param[{net}] {k}", sd[k], st.params[net][k], steps, g.hp.lr)
        for k in ("feat_ss.weight", "feat_ss2.weight", "feat_ss3.bias"):
            assert torch.equal(sd[k].cpu(), st.params[net][k])         # dead tensors never move
    for i in range(2):
        report_after_updates(f"bank{i} feats", eng.bank_feats[i], st.bank_feats[i], 1e-5, 5e-6, steps, g.hp.lr)
        report(f"bank{i} probs", eng.bank_probs[i], st.bank_probs[i], 1e-4, 2e-4)   # softmax of |logits|~100 (peaky case)


@pytest.mark.parametrize("name", ["b2_64", "b2_b256", "b4_b256", "p_b256_ep1"])
def test_first_step_gradients_with_the_oracles_own_relu_decisions(name):
    """The step test above hands the DEVICE's ReLU decisions to the oracle.  Here the oracle keeps its OWN sign(z)
    decisions for every element except those whose pre-activation lies within 2e-5 of zero (where fp32 summation order
    decides, and only there the device's decision is taken): every gradient tensor of the first step of a reference
    fixture must still meet the tight elementwise bound -- at the headline batch (B2, 128 + 128), at 200 bands (B4) and
    at the reference's own shape (P).  Decisions that differ OUTSIDE that band are counted for all three ReLUs (conv1,
    conv2, spectral) and must be zero; pixels the floor-pooling drops carry no gradient and no device decision."""
    g = GoldenCase(name)
    eng, p0, p1 = _engine(g)
    st = O.StepState.create(g.shape, p0, p1, g.bt, g.hp)
    n = g.bt + g.btu
    b = g.batch(0)
    epoch, bi = g.epoch_bi(0)
    cb = cuda_batch(b)
    eng.step(cb["XPl"], cb["Xl"], cb["Y"], cb["XPu"], cb["Xu"], epoch, bi, noise=cb["noise"], dropmask=cb["dropmask"])
    dev_gates = hip_relu_gates(eng, g.shape, n)
    import copy
    probe = O.train_step(copy.deepcopy(st), b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"],
                         epoch, bi, g.hp)                                   # own decisions: only its pre-activations are used
    H2, W2 = g.shape.H // 2, g.shape.W // 2
    gates, nbound, ndiff, nel = [], 0, {"z1": 0, "z2": 0, "zy": 0}, 0
    for net in range(2):
        gn = {}
        for key in ("z1", "z2", "zy"):
            zo = probe["taps"][net][key]
            own = zo > 0
            dev = dev_gates[net][key].reshape(own.shape)
            boundary = zo.abs() < 2e-5
            live = torch.ones_like(own)                      # elements the device computes (the pooled windows)
            if key == "z1":
                live[:, :, 2 * H2:, :] = False; live[:, :, :, 2 * W2:] = False
            elif key == "z2":
                live[:, :, 2 * (H2 // 2):, :] = False; live[:, :, :, 2 * (W2 // 2):] = False
            nbound += int((boundary & live).sum())
            nel += int(live.sum())
            ndiff[key] += int(((own != dev) & ~boundary & live).sum())
            gn[key] = torch.where(boundary & live, dev, own)
        gates.append(gn)
    print(f"[{name}] {nel} ReLU elements, {nbound} within 2e-5 of the boundary; decisions differing elsewhere: {ndiff}")
    assert sum(ndiff.values()) == 0, ndiff
    ref = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"], epoch, bi, g.hp,
                       relu_gates=gates)
    for net in range(2):
        for k in O.LIVE_KEYS:
            gr = ref["grads"][net][k]
            mx = max(float(gr.abs().max()), 1e-4)
            report(f"grad[{net}] {k}", eng.grad(net, k), gr, 5e-4, 5e-5 * mx)


def test_b5_1to8_single_gpu_runs_on_with_modulo_bank_writes():
    """BASELINE configs[4]'s batch split (64 labelled + 512 unlabelled, 15x15x48, 20 classes) on ONE GPU through
    TrainEngine.  The reference runs exactly one step at this split (its second bank slice-assign raises):
    step 0 is held to the reference's own numbers (fixture b5_1to8); the steps behind it write the banks modulo Q
    (Q = 640 < 576 + 256: every write overlaps the previous step's rows and wraps) and are held to the oracle's
    modulo write -- losses, counts, every gradient, pointers, banks."""
    g = GoldenCase("b5_1to8")
    eng, p0, p1 = _engine(g)
    st = O.StepState.create(g.shape, p0, p1, g.bt, g.hp)
    n, steps = g.bt + g.btu, 4
    for s in range(steps):
        b = g.batch(s)
        epoch, bi = g.epoch_bi(s)
        cb = cuda_batch(b)
        eng.step(cb["XPl"], cb["Xl"], cb["Y"], cb["XPu"], cb["Xu"], epoch, bi, noise=cb["noise"],
                 dropmask=cb["dropmask"])
        gates = hip_relu_gates(eng, g.shape, n)
        ref = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"],
                           epoch, bi, g.hp, relu_gates=gates)
        sc = eng.read_scalars()
        row = [sc[k] for k in ("ctr_s", "total_s", "cls_s", "con_s", "acc")]
        print(f"[b5_1to8 x{steps}] step {s}: hip={row} oracle={ref['hist']}")
        if s == 0:
            assert rel_err(row, g.z["hist"][0], 1e-7) < LOSS_RTOL, (row, g.z["hist"][0])     # the reference itself
        assert rel_err(row, ref["hist"], 1e-7) < LOSS_RTOL, (s, row, ref["hist"])
        assert eng.ptr == list(st.ptr), (s, eng.ptr, st.ptr)
        relu_mask_audit(eng, ref["taps"], g.shape, n, ztol=2e-5 + 0.1 * g.hp.lr * s, ztol_y=2e-5 + 0.25 * g.hp.lr * s)
        for net in range(2):
            for k in O.LIVE_KEYS:
                gr = ref["grads"][net][k]
                report(f"grad[{net}] {k}", eng.grad(net, k), gr, 5e-4, 5e-5 * max(float(gr.abs().max()), 1e-4))
    assert eng.Q == 640 and eng.ptr == [(steps * 256) % 640, (steps * 256 + 256) % 640]
    for i in range(2):
        report_after_updates(f"bank{i} feats", eng.bank_feats[i], st.bank_feats[i], 1e-5, 5e-6, steps, g.hp.lr)
        report(f"bank{i} probs", eng.bank_probs[i], st.bank_probs[i], 1e-4, 2e-4)


@pytest.mark.parametrize("name", NAN_CASES)
def test_dead_relu_rows_propagate_nan_like_the_reference(name):
    """SURVEY.md section 4 regime (v): a sample whose spectral ReLU output is all zero makes Normalize
    (tools/models.py:87-90, no epsilon) return 0/0 = NaN.  The reference then carries NaN through the similarity
    matrices, the smoothed probabilities, both contrastive and both mutual losses, EVERY gradient element and two
    rows of each bank; CE and accuracy of the first step stay finite.  The HIP path must land NaN on exactly the
    same places and keep the finite values."""
    g = GoldenCase(name)
    eng, p0, p1 = _engine(g)
    st = O.StepState.create(g.shape, p0, p1, g.bt, g.hp)
    z = g.z
    for s in range(g.steps):
        b = g.batch(s)
        epoch, bi = g.epoch_bi(s)
        cb = cuda_batch(b)
        ref = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"], epoch, bi, g.hp)
        eng.step(cb["XPl"], cb["Xl"], cb["Y"], cb["XPu"], cb["Xu"], epoch, bi, noise=cb["noise"],
                 dropmask=cb["dropmask"])
        sc = eng.read_scalars()
        row = [sc[k] for k in ("ctr_s", "total_s", "cls_s", "con_s", "acc")]
        extra = [sc[k] for k in ("total_w", "cls_w", "con_w", "ctr_w")]
        print(f"[{name}] step {s}: hip={row} golden={list(z['hist'][s])}")
        assert rel_err(row, z["hist"][s], 1e-7) < LOSS_RTOL, (s, row, z["hist"][s])        # NaN at the same places
        assert rel_err(extra, z["extra"][s], 1e-7) < LOSS_RTOL, (s, extra, z["extra"][s])
        assert [sc["n_mask_w"], sc["n_mask_s"], sc["n_pos"], sc["n_neg"]] == list(z["counts"][s])
        lo, fe = eng.outputs()
        if s == 0:
            report("logits", lo, torch.stack(ref["logits"]), 2e-4, 5e-5)                      # finite in step 0
            fr = torch.stack(ref["feats"])
            assert torch.equal(torch.isnan(fe).cpu(), torch.isnan(fr))
            assert int(torch.isnan(fr).sum()) == 2 * 2 * 1024                                  # 2 dead rows per net
            ok = ~torch.isnan(fr)
            report("feat (live rows)", fe.cpu()[ok], fr[ok], 1e-5, 3e-6)
        for net in range(2):
            got = [int(torch.isnan(eng.grad(net, k)).sum()) for k in O.LIVE_KEYS]
            assert got == [int(v) for v in z["grad_nan"][s][net]], (s, net, got)
        got = [int(torch.isnan(t).sum()) for t in (eng.bank_feats[0], eng.bank_probs[0], eng.bank_feats[1],
                                                   eng.bank_probs[1])]
        assert got == [int(v) for v in z["bank_nan"][s]], (s, got, z["bank_nan"][s])
        if s == 0:      # the finite part of the banks is still exact
            for i in range(2):
                fin = ~torch.isnan(st.bank_feats[i])
                report(f"bank{i} feats (finite part)", eng.bank_feats[i].cpu()[fin], st.bank_feats[i][fin], 1e-5, 5e-6)


def test_full_batch_properties_b2():
    """BASELINE configs[1] size (11x11x103, 128+128), Philox noise/dropout: size-independent checks."""
    from cmlpl_amd import TrainEngine
    shape = O.NetShape(103, 11, 11, 103, 9)
    hp = O.HyperParams()
    b = cuda_batch(O.synthetic_batch(shape, 128, 128, 77, with_noise=False))

    def run(seed, steps):
        eng = TrainEngine(to_shape(shape), 128, 128, to_hp(hp), device=DEV, seed=seed)
        eng.load_state_dict(0, O.closed_form_params(shape, 1))
        eng.load_state_dict(1, O.closed_form_params(shape, 2))
        rows = []
        for s in range(steps):
            eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, s)
            rows.append(eng.read_scalars())
        return eng, rows

    e1, r1 = run(5, 6)
    e2, r2 = run(5, 6)
    e3, r3 = run(6, 6)
    for a, c in zip(r1, r2):          # same seed -> bit-identical trajectory (deterministic kernels)
        assert a == c
    assert r1[0]["total_s"] != r3[0]["total_s"]                       # different Philox stream
    assert all(np.isfinite(list(r.values())).all() for r in r1)
    assert r1[-1]["cls_s"] < r1[0]["cls_s"]                            # same batch re-fit: CE falls
    assert e1.ptr == [(6 * 256) % 1280, (6 * 256 + 256) % 1280]       # train.py:234,237
    lo, fe = e1.outputs()
    nrm = fe.norm(dim=2)
    assert torch.allclose(nrm, torch.ones_like(nrm), atol=1e-5)       # Normalize: unit rows
    # bank rows written this step are exactly the features of this step (copy, bit-exact)
    p0 = (5 * 256) % 1280
    assert torch.equal(e1.bank_feats[0][p0:p0 + 128], fe[1][128:])     # fU_w
    assert torch.equal(e1.bank_feats[0][p0 + 128:p0 + 256], fe[0][:128])  # fL_s
    onehot = torch.zeros(128, 9, device=DEV).scatter(1, b["Y"].view(-1, 1), 1)
    assert torch.equal(e1.bank_probs[1][(p0 + 256) % 1280 + 128:(p0 + 256) % 1280 + 256], onehot)


def test_philox_noise_statistics():
    """augment kernel in Philox mode: N(0, sigma^2) per element, independent per network."""
    import ctypes as C
    from cmlpl_amd import _lib
    lib = _lib.load()
    shape = O.NetShape(103, 11, 11, 103, 9)
    cs = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
    bt = btu = 64
    xpl = torch.zeros(bt, 103, 11, 11, device=DEV); xpu = torch.ones(btu, 103, 11, 11, device=DEV)
    xl = torch.zeros(bt, 103, device=DEV); xu = torch.ones(btu, 103, device=DEV)
    xn = torch.empty(2, bt + btu, 103 * 121, device=DEV); sn = torch.empty(2, bt + btu, 103, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.cmlpl_augment(C.byref(cs), 2, bt, btu, xpl.data_ptr(), xl.data_ptr(), xpu.data_ptr(), xu.data_ptr(),
                             None, 0.5, 123, 7, None, xn.data_ptr(), sn.data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    lab, unl = xn[:, :bt], xn[:, bt:] - 1.0
    for t in (lab, unl):
        assert abs(float(t.mean())) < 2e-3 and abs(float(t.std()) - 0.5) < 2e-3
    z = torch.cat([lab.flatten(), unl.flatten()]) / 0.5
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05                    # Gaussian kurtosis
    c = torch.corrcoef(torch.stack([xn[0, :bt].flatten(), xn[1, :bt].flatten()]))[0, 1]
    assert abs(float(c)) < 5e-3                                         # nets draw independent noise
    assert abs(float(sn[:, :bt].std()) - 0.5) < 1e-2
    xn2 = torch.empty_like(xn)
    assert lib.cmlpl_augment(C.byref(cs), 2, bt, btu, xpl.data_ptr(), xl.data_ptr(), xpu.data_ptr(), xu.data_ptr(),
                             None, 0.5, 123, 7, None, xn2.data_ptr(), sn.data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(xn, xn2)                                          # counter-based: reproducible


def test_device_noise_is_the_documented_generator():
    """The augmentation noise of the kernels, value by value, against the numpy restatement of pcg4d + Box-Muller in
    tests/test_noise_generator_math.py: zero inputs and sigma = 1 make cmlpl_augment return the noise itself.  Patches:
    noise_normal8 (eight normals per hash call, counter = (global sample << 24) | pair of 16-byte groups, stream 0x100 +
    network); spectra: noise_normal4 (counter = .. | group, stream 0x200 + network).  The fused forward forms the SAME
    patch values (it leaves the rows it augmented in the workspace: checked below through a training step with zero
    patches).  Tolerance 2e-5 absolute: the kernel takes log2 / sin / cos from the hardware's approximations."""
    import ctypes as C
    from cmlpl_amd import _lib
    from tests.test_noise_generator_math import noise_normal4, noise_normal8, _ctr
    lib = _lib.load()
    shape = O.NetShape(103, 11, 11, 103, 9)
    cs = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
    bt = btu = 4
    per = 103 * 121
    xpl = torch.zeros(bt, 103, 11, 11, device=DEV); xpu = torch.zeros(btu, 103, 11, 11, device=DEV)
    xl = torch.zeros(bt, 103, device=DEV); xu = torch.zeros(btu, 103, device=DEV)
    xn = torch.empty(2, bt + btu, per, device=DEV); sn = torch.empty(2, bt + btu, 103, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    seed, step = 1088, 3
    assert lib.cmlpl_augment(C.byref(cs), 2, bt, btu, xpl.data_ptr(), xl.data_ptr(), xpu.data_ptr(), xu.data_ptr(),
                             None, 1.0, seed, step, None, xn.data_ptr(), sn.data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    pairs = np.arange((per + 7) // 8)
    sgroups = np.arange((103 + 3) // 4)
    for net in range(2):
        for row in (0, 3, 5):                      # rows 0..3 labelled (global sample = row), 4..7 unlabelled (bt + i)
            want = noise_normal8(seed, step, 0x100 + net, _ctr(row, pairs)).T.reshape(-1)[:per]
            got = xn[net, row].cpu().numpy()
            assert np.abs(got - want).max() < 2e-5, (net, row, np.abs(got - want).max())
            wants = noise_normal4(seed, step, 0x200 + net, _ctr(row, sgroups)).T.reshape(-1)[:103]
            assert np.abs(sn[net, row].cpu().numpy() - wants).max() < 2e-5
    # the fused per-sample forward forms the SAME patch values: zero patches and sigma = 1 -> the rows it keeps for the
    # backward ARE its noise (160 + 160 rows: the four-wave kernels; 24 + 24: the eight-wave ones, one workgroup per CU)
    from cmlpl_amd import HyperParams, NetShape, TrainEngine
    for bt2 in (160, 24):
        eng = TrainEngine(NetShape(103, 11, 11, 103, 9), bt2, bt2, HyperParams(noise=1.0), device=DEV, seed=seed)
        eng.init_params_default(1)
        z = lambda *sh: torch.zeros(*sh, device=DEV)
        eng.step(z(bt2, 103, 11, 11), z(bt2, 103), torch.zeros(bt2, dtype=torch.int64, device=DEV), z(bt2, 103, 11, 11),
                 z(bt2, 103), 0, 0)
        torch.cuda.synchronize()
        rows = eng.debug_region("xn").view(2, 2 * bt2, per)
        for net in range(2):
            for row in (0, bt2 - 1, bt2, 2 * bt2 - 1):          # global sample = row (labelled rows first, then bt + i)
                want = noise_normal8(seed, 0, 0x100 + net, _ctr(row, pairs)).T.reshape(-1)[:per]
                assert np.abs(rows[net, row].cpu().numpy() - want).max() < 2e-5, (bt2, net, row)


@pytest.mark.parametrize("win,C", [(9, 30), (8, 103)])
def test_step_by_index_where_forward_and_backward_plans_differ(win, C):
    """A 9x9 window at 128 + 128 rows plans an UNFUSED forward (two samples per conv1 workgroup) and a FUSED data
    gradient: conv0's weight gradient must then come from the rows the forward saw (augmented, taken through the index
    lists) -- the fused data gradient reads plain rows by batch row, so it has to be handed the augmented copy in the
    workspace, never the raw batch (round 4 handed it the raw batch).  Batches by index into larger resident splits,
    explicit noise on, every gradient against the oracle on the gathered rows; 8x8 (fused both ways) rides along."""
    from cmlpl_amd import TrainEngine
    shape = O.NetShape(C, win, win, 103, 9)
    bt = btu = 128
    n = bt + btu
    hp = O.HyperParams()
    p0, p1 = O.closed_form_params(shape, 51), O.closed_form_params(shape, 52)
    eng = TrainEngine(to_shape(shape), bt, btu, to_hp(hp), device=DEV)
    eng.load_state_dict(0, p0); eng.load_state_dict(1, p1)
    st = O.StepState.create(shape, p0, p1, bt, hp)
    g = torch.Generator().manual_seed(9)
    NL, NU = 200, 333
    for s in range(2):
        b = O.synthetic_batch(shape, bt, btu, 7700 + s)
        li, ui = torch.randperm(NL, generator=g)[:bt], torch.randperm(NU, generator=g)[:btu]
        # resident splits: the batch rows scattered to positions li / ui, other rows filled with junk the step must not see
        XPl = torch.full((NL, C, win, win), 7.0); Xl = torch.full((NL, 103), -5.0); Y = torch.zeros(NL, dtype=torch.int64)
        XPu = torch.full((NU, C, win, win), -9.0); Xu = torch.full((NU, 103), 3.0)
        XPl[li], Xl[li], Y[li] = b["XPl"], b["Xl"], b["Y"]
        XPu[ui], Xu[ui] = b["XPu"], b["Xu"]
        cb = cuda_batch(b)
        eng.step(XPl.to(DEV), Xl.to(DEV), Y.to(DEV), XPu.to(DEV), Xu.to(DEV), 1, s, noise=cb["noise"],
                 dropmask=cb["dropmask"], lab_idx=li.to(DEV), unl_idx=ui.to(DEV))
        gates = hip_relu_gates(eng, shape, n)
        ref = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"], 1, s, hp,
                           relu_gates=gates)
        sc = eng.read_scalars()
        for k in ("ctr_s", "total_s", "cls_s", "con_s", "total_w", "cls_w", "con_w", "ctr_w"):
            want = float(ref[k].detach())
            assert abs(sc[k] - want) <= LOSS_RTOL * abs(want) + 1e-6, (s, k, sc[k], want)
        relu_mask_audit(eng, ref["taps"], shape, n, ztol=2e-5 + 0.1 * hp.lr * s, ztol_y=2e-5 + 0.25 * hp.lr * s)
        for net in range(2):
            for k in O.LIVE_KEYS:
                gr = ref["grads"][net][k]
                mx = max(float(gr.abs().max()), 1e-4)
                report(f"[{win}x{win}] step {s} grad[{net}] {k}", eng.grad(net, k), gr, 5e-4, 5e-5 * mx)
