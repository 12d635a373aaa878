"""GPU: whole-image inference straight from the scene cube (cmlpl_infer_cube; SURVEY.md 8f N1 x N3 joined: reference
tools/hyper_tools.py:416-437 test_whole over the windows of :226-243 ExtractPatches, train.py:291-294) against the
oracle's extract_patches -> basenet2_forward (eval) -> argmax, and against the library's own patch path."""
import numpy as np
import pytest
import torch

from oracle import cmlpl_oracle as O
from tests.gpu_util import DEV, report

pytestmark = pytest.mark.gpu


def _scene(rows, cols, C, bands, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    cube = rng.standard_normal((rows, cols, C)).astype(np.float32)
    X = rng.standard_normal((rows * cols, bands)).astype(np.float32)
    return cube, X


def _module(shape, seed, scale=1.0):
    from cmlpl_amd.models import BaseNet2
    p = O.closed_form_params(shape, seed)
    p["classifier.weight"] = p["classifier.weight"] * scale
    net = BaseNet2(num_features=shape.bands, dropout=0.8, num_classes=shape.K, in_channels=shape.C, window=shape.H).to(DEV)
    net.load_state_dict(p)
    net.eval()
    return net, p


@pytest.mark.parametrize("name,shape,rows,cols", [("B2", (103, 11, 11, 103, 9), 64, 48), ("B4", (200, 11, 11, 200, 16), 20, 24),
                                                  ("B5", (48, 15, 15, 48, 20), 24, 40), ("W8", (30, 8, 8, 30, 5), 16, 16),
                                                  ("W12", (16, 12, 12, 16, 7), 14, 19), ("P", (60, 20, 20, 103, 9), 22, 26)])
def test_infer_cube_matches_the_oracle(name, shape, rows, cols):
    """labels equal the oracle's extract_patches -> basenet2_forward -> argmax wherever the top two logits are further
    apart than the logits tolerance; logits within 1e-4 (relative to the row's largest); every pixel of the scene, i.e.
    every kind of mirrored border.  P = the reference's own 20 x 20 x 60 window (tools/models.py:102,127): more pixels
    than the fused forward takes, so infer_cube cuts the windows of a chunk on the device and runs the general forward."""
    from cmlpl_amd.infer import infer_cube
    s = O.NetShape(*shape)
    cube, X = _scene(rows, cols, s.C, s.bands, 99)
    net, p = _module(s, 61, scale=8.0)
    want = []
    with torch.no_grad():
        for o in range(0, rows * cols, 512):
            idx = np.arange(o, min(o + 512, rows * cols))
            XP = torch.from_numpy(O.extract_patches(cube, s.H, idx))
            z, _ = O.basenet2_forward(p, XP, torch.from_numpy(X[idx]), None)
            want.append(z)
    want = torch.cat(want)
    labels, logits = infer_cube(net, torch.from_numpy(cube).to(DEV), torch.from_numpy(X).to(DEV), chunk=1000, want_logits=True)
    tol = 1e-4 * float(want.abs().max()) + 1e-5
    report(f"[{name}] logits", logits, want, 0.0, tol)
    top2 = want.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 4 * tol
    assert int(sure.sum()) > 0.9 * len(sure)
    assert torch.equal(labels.cpu()[sure], want.argmax(1)[sure]), "labels differ away from ties"
    # a sub-range that starts and ends off the 8-pixel groups of the XCD dealing
    lab2 = infer_cube(net, torch.from_numpy(cube).to(DEV), torch.from_numpy(X).to(DEV), pixel0=5, n=rows * cols - 16)
    assert torch.equal(lab2, labels[5:rows * cols - 11])


def test_infer_cube_equals_the_patch_path_and_test_whole_takes_a_cube():
    """cube path vs this library's own patches -> BaseNet2.forward (eval) path: same labels, logits to rounding;
    tools.hyper_tools.test_whole gives the same prediction from a CubeSource as from a DataLoader of patches"""
    from cmlpl_amd.infer import CubeSource, infer_cube
    from cmlpl_amd.patches import extract_patches
    from tools.hyper_tools import test_whole
    s = O.NetShape(103, 11, 11, 103, 9)
    rows, cols = 40, 33
    cube, X = _scene(rows, cols, s.C, s.bands, 5)
    net, _ = _module(s, 13, scale=8.0)
    dc, dx = torch.from_numpy(cube).to(DEV), torch.from_numpy(X).to(DEV)
    labels, logits = infer_cube(net, dc, dx, want_logits=True)
    XP = extract_patches(dc, torch.arange(rows * cols, device=DEV), s.H)
    with torch.no_grad():
        z, _ = net(XP, dx)
    report("logits cube vs patches", logits, z, 1e-5, 1e-5 * float(z.abs().max()))
    top2 = z.topk(2, dim=1).values
    sure = ((top2[:, 0] - top2[:, 1]) > 1e-4 * float(z.abs().max())).cpu()
    assert torch.equal(labels.cpu()[sure], z.argmax(1).cpu()[sure])
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(XP.cpu(), dx.cpu()), batch_size=512)
    a = test_whole(net, loader, print_per_batches=10 ** 9)
    b = test_whole(net, CubeSource(dc, dx))
    assert a.shape == b.shape == (rows * cols,)
    assert np.array_equal(a[sure.numpy()], b[sure.numpy()])


def test_infer_cube_refuses_what_it_cannot_take():
    from cmlpl_amd.infer import infer_cube, infer_fused, infer_supported
    s = O.NetShape(60, 20, 20, 103, 9)                   # the reference's own 20 x 20 window: 400 pixels
    assert not infer_fused(s) and infer_fused(O.NetShape(103, 11, 11, 103, 9)) and infer_fused(O.NetShape(48, 15, 15, 48, 20))
    assert infer_supported(s) and infer_supported(O.NetShape(103, 11, 11, 103, 9))      # (by chunks of extracted windows)
    cube, X = _scene(24, 24, 60, 103, 1)
    net2, _ = _module(O.NetShape(103, 11, 11, 103, 9), 3)
    with pytest.raises(ValueError):
        infer_cube(net2, torch.from_numpy(cube).to(DEV), torch.from_numpy(X).to(DEV))      # 60 channels into a 103-channel net
