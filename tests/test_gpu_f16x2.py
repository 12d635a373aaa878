"""GPU: conv1's tap loops of the per-sample kernels on TWO fp16 pieces (csrc/conv3x3.hip: conv3_taps_ks_h; DESIGN.md
section 4) -- that the path is the one that runs, how close it stays to the three-piece bf16 loop, and that every sample /
network whose operands leave the ranges the scheme needs falls back to that loop BIT FOR BIT: weights beyond fp16 at
the packing scale (set at load time, or reached by an optimizer step: the flag is sticky until the next full pack), an
image too large, too small or all zero.  (Parity against the oracle / the reference fixture / the per-element bounds runs
on the default path everywhere else in the suite, and on the three-piece path in tests/test_gpu_env_paths.py.)"""
import os

import pytest
import torch

from oracle import cmlpl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = {"B2": O.NetShape(103, 11, 11, 103, 9), "B4": O.NetShape(200, 11, 11, 200, 16)}


def _run(shape, params, bt, btu, scale=1.0, steps=1, lr=None, f16x2="1", noise=None, env=None):
    """`steps` training steps from `params` (both networks) under CMLPL_F16X2 = f16x2 -> (logits, grads, packed flag words)"""
    from cmlpl_amd import HyperParams, NetShape, TrainEngine, _lib
    lib = _lib.load()
    old = os.environ.get("CMLPL_F16X2")
    os.environ["CMLPL_F16X2"] = f16x2
    old_env = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    lib.cmlpl_debug_reload_switches()
    try:
        hp = HyperParams(**{k: v for k, v in (("lr", lr), ("noise", noise)) if v is not None})
        eng = TrainEngine(NetShape(shape.C, shape.H, shape.W, shape.bands, shape.K), bt, btu, hp, device=DEV, seed=3)
        for net in range(2):
            eng.load_state_dict(net, params)
        b = O.synthetic_batch(shape, bt, btu, 41)
        d = lambda t: (t * scale if t.is_floating_point() else t).to(DEV)
        for s in range(steps):
            eng.step(d(b["XPl"]), d(b["Xl"]), d(b["Y"]), d(b["XPu"]), d(b["Xu"]), 1, s, apply_update=(s + 1 < steps))
        torch.cuda.synchronize()
        import ctypes as C
        from cmlpl_amd import _lib as L
        cs = L.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
        lay = L.layout(cs)
        flag_off = int(lay.packed_total) - 16
        flags = eng.packed.view(2, -1)[:, flag_off].view(torch.int32).cpu().tolist()
        return eng.logits.clone().cpu(), eng.grads.clone().cpu(), flags
    finally:
        if old is None:
            os.environ.pop("CMLPL_F16X2", None)
        else:
            os.environ["CMLPL_F16X2"] = old
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.cmlpl_debug_reload_switches()


@pytest.mark.parametrize("name,bt,btu", [("B2", 16, 16), ("B4", 8, 8), ("B2", 4, 4)], ids=["B2-four-waves", "B4", "B2-eight-waves"])
def test_two_piece_path_runs_and_stays_at_fp32_level(name, bt, btu):
    shape = SHAPES[name]
    params = O.closed_form_params(shape, 9)
    env8 = os.environ.get("CMLPL_KS8")
    if bt == 4:
        os.environ["CMLPL_KS8"] = "1"            # the eight-wave kernels (what a rank's shard launches)
    try:
        lo1, g1, f1 = _run(shape, params, bt, btu, f16x2="1")
        lo0, g0, f0 = _run(shape, params, bt, btu, f16x2="0")
    finally:
        if env8 is None:
            os.environ.pop("CMLPL_KS8", None)
        else:
            os.environ["CMLPL_KS8"] = env8
    assert f1 == [0, 0] and f0 == [0, 0]
    assert not torch.equal(lo1, lo0) and not torch.equal(g1, g0), "the two-piece loop did not run"
    # two roundings of the same fp32-level quantity: the loops differ by a few ulp of the largest element
    assert (lo1 - lo0).abs().max() <= 4e-6 * lo0.abs().max(), ((lo1 - lo0).abs().max(), lo0.abs().max())
    assert (g1 - g0).abs().max() <= 4e-6 * g0.abs().max(), ((g1 - g0).abs().max(), g0.abs().max())


def test_weights_beyond_fp16_at_the_packing_scale_fall_back_bit_for_bit():
    shape = SHAPES["B2"]
    params = O.closed_form_params(shape, 9)
    big = {k: v.clone() for k, v in params.items()}
    big["conv1.weight"][3, 5, 1, 1] = 9.0              # 9 * 2^13 > 65504
    lo1, g1, f1 = _run(shape, big, 8, 8, f16x2="1")
    lo0, g0, _ = _run(shape, big, 8, 8, f16x2="0")
    assert f1 == [1, 1]
    assert torch.equal(lo1, lo0) and torch.equal(g1, g0)
    # ... and a full pack of ordinary weights clears the flag again (a fresh engine packs at load time)
    _, _, f2 = _run(shape, params, 8, 8, f16x2="1")
    assert f2 == [0, 0]


def test_an_optimizer_step_that_leaves_the_range_raises_the_flag():
    """Adam's re-packing sets the flag itself: weights at 7.9 that a step of lr = 0.5 carries past 8 -- the NEXT forward
    (which reads the sets that step wrote) already runs the three-piece loop: two steps equal the three-piece run bitwise
    except for what the first step's two-piece forward / backward left in the parameters, so compare the flag and that
    a third-step forward agrees with a three-piece engine started from the same updated parameters."""
    shape = SHAPES["B2"]
    params = O.closed_form_params(shape, 9)
    near = {k: v.clone() for k, v in params.items()}
    near["conv1.weight"][:, :, 1, 1] = 7.9 * torch.sign(near["conv1.weight"][:, :, 1, 1] + 1e-12)
    _, _, f_before = _run(shape, near, 8, 8, steps=1, lr=0.5, f16x2="1")        # (no update applied in a one-step run)
    _, _, f_after = _run(shape, near, 8, 8, steps=2, lr=0.5, f16x2="1")         # one Adam step in between
    assert f_before == [0, 0] and f_after == [1, 1], (f_before, f_after)


@pytest.mark.parametrize("scale", [1e30, 1e-30, 0.0], ids=["huge", "tiny", "zero"])
def test_images_outside_the_scale_range_fall_back_bit_for_bit(scale):
    """inputs scaled so that conv0's output (the image conv1 reads) is ~1e30 / ~1e-30 x / exactly its bias: the forward
    kernel's per-sample exponent check sends the first two to the three-piece loop; with zero input the image is the
    bias pattern (ordinary numbers: the two-piece loop runs, results at fp32 level)"""
    shape = SHAPES["B2"]
    params = O.closed_form_params(shape, 9)
    if scale == 1e-30:                                 # (the bias would dominate: remove it so that the image itself is tiny)
        params = {k: (torch.zeros_like(v) if k == "conv0.bias" else v.clone()) for k, v in params.items()}
    # (no augmentation noise: it would be the image)
    lo1, g1, _ = _run(shape, params, 8, 8, scale=scale, f16x2="2", noise=0.0)   # forward kernel only
    lo0, g0, _ = _run(shape, params, 8, 8, scale=scale, f16x2="0", noise=0.0)
    if scale == 0.0:
        assert torch.isfinite(lo1).all() and (lo1 - lo0).abs().max() <= 4e-6 * lo0.abs().max().clamp_min(1e-30)
    else:
        both_nan = torch.isnan(lo1) & torch.isnan(lo0)
        assert torch.equal(torch.where(both_nan, torch.zeros_like(lo1), lo1), torch.where(both_nan, torch.zeros_like(lo0), lo0))


def test_the_library_reports_which_products_run_on_two_pieces():
    """cmlpl_debug_two_piece (bench.py's roofline mix): per-sample kernels -> conv1's two tap loops + the weight gradients;
    the general path (the reference's 20 x 20 window) -> also conv2's launches; nothing with the switch off"""
    import ctypes as C
    from cmlpl_amd import _lib
    lib = _lib.load()
    ask = lambda shape, n: lib.cmlpl_debug_two_piece(C.byref(_lib.Shape(*shape)), 2, n)
    old = os.environ.get("CMLPL_F16X2")
    try:
        os.environ.pop("CMLPL_F16X2", None)
        lib.cmlpl_debug_reload_switches()
        assert ask((103, 11, 11, 103, 9), 256) == 7 and ask((103, 11, 11, 103, 9), 128) == 7      # four waves | eight waves
        assert ask((48, 15, 15, 48, 20), 256) == 7                                                 # eight-tile kernels
        assert ask((60, 20, 20, 103, 9), 256) == 15                                                # general path
        os.environ["CMLPL_F16X2"] = "2"
        lib.cmlpl_debug_reload_switches()
        assert ask((103, 11, 11, 103, 9), 256) == 1
        os.environ["CMLPL_F16X2"] = "0"
        lib.cmlpl_debug_reload_switches()
        assert ask((103, 11, 11, 103, 9), 256) == 0 and ask((60, 20, 20, 103, 9), 256) == 0
    finally:
        if old is None:
            os.environ.pop("CMLPL_F16X2", None)
        else:
            os.environ["CMLPL_F16X2"] = old
        lib.cmlpl_debug_reload_switches()


def test_general_path_two_piece_loops_run_and_stay_at_fp32_level():
    """the reference's own 20 x 20 x 60 window: all four 3x3 launches and the weight gradients on two pieces against the
    three-piece run of the same step"""
    shape = O.NetShape(60, 20, 20, 103, 9)
    params = O.closed_form_params(shape, 9)
    lo1, g1, f1 = _run(shape, params, 8, 8, f16x2="1")
    lo0, g0, _ = _run(shape, params, 8, 8, f16x2="0")
    assert f1 == [0, 0]
    assert not torch.equal(lo1, lo0) and not torch.equal(g1, g0), "the two-piece loops did not run"
    assert (lo1 - lo0).abs().max() <= 4e-6 * lo0.abs().max(), ((lo1 - lo0).abs().max(), lo0.abs().max())
    assert (g1 - g0).abs().max() <= 4e-6 * g0.abs().max(), ((g1 - g0).abs().max(), g0.abs().max())


@pytest.mark.parametrize("name", ["B2", "P"])
def test_weight_gradients_skip_samples_without_gradient(name):
    """At the default threshold no unlabelled row of a freshly initialised pair of networks is confident (train.py:221):
    their gradient images are zero everywhere and the two-piece weight-gradient launch leaves those samples out
    (CMLPL_ZERO_SKIP=0: walks them) -- the same sums, grouped differently: equal to fp32 rounding; so do the fused
    backward's workgroup pairing (CMLPL_BWD_PAIR) and zero-image skip."""
    shape = SHAPES[name] if name in SHAPES else O.NetShape(60, 20, 20, 103, 9)
    params = O.closed_form_params(shape, 9)
    lo1, g1, _ = _run(shape, params, 24, 40)
    lo0, g0, _ = _run(shape, params, 24, 40, env={"CMLPL_ZERO_SKIP": "0", "CMLPL_BWD_PAIR": "0"})
    assert torch.equal(lo1, lo0)                              # (the forward is not touched)
    assert (g1 - g0).abs().max() <= 2e-6 * g0.abs().max(), ((g1 - g0).abs().max(), g0.abs().max())
    if name == "B2":
        assert not torch.equal(g1, g0), "no sample was skipped: were the unlabelled rows confident?"
