"""Whole-image inference straight from the scene cube (reference tools/hyper_tools.py:416-437 ``test_whole`` over the
patches of :226-243 ``ExtractPatches``; train.py:291-294).  The reference materialises every pixel's window (19.9 GB
for PaviaU) and streams it through a DataLoader; here the z-scored / PCA'd cube [rows, cols, C] and the spectra
[rows * cols, bands] stay resident in HBM and ``cmlpl_infer_cube`` gathers each window through the mirror index while
the fused forward stages its input slab -- no patch tensor exists, the argmax is written on the device."""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch

from . import _lib


class CubeSource(NamedTuple):
    """what ``tools.hyper_tools.test_whole`` takes instead of a DataLoader of patches"""
    cube: torch.Tensor          # [rows, cols, C] float32 cuda, band-last (the scene the patches are cut from)
    spectra: torch.Tensor       # [rows * cols, bands] float32 cuda (row-major pixel order)


def _net_buffers(net):
    """(shape struct, flat parameters, packed weights) of ONE network: a cmlpl_amd.models.BaseNet2 module, or a
    (TrainEngine, net index) pair."""
    if isinstance(net, tuple):
        eng, i = net
        eng._ensure_packed(C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream))
        return eng.cshape, eng.params[i], eng.packed[i]
    flat, packed = net._flat_params(net._live_params())
    return net._cshape, flat, packed


def infer_fused(shape) -> bool:
    """does ``cmlpl_infer_cube`` -- the fused eval forward that gathers the window while it stages its slab -- take this
    window shape?  (square windows of 8 x 8 up to 256 pixels whose final pooled map the per-sample fused forward covers;
    asked of the library: its workspace size is 0 otherwise)"""
    lib = _lib.load()
    cs = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
    return lib.cmlpl_infer_workspace_bytes(C.byref(cs), 8) > 0


def infer_supported(shape) -> bool:
    """can ``infer_cube`` label a scene of this window shape straight from its cube?  Every square window the network
    runs on: the fused kernel where it applies (``infer_fused``), else -- the reference's own 20 x 20 x 60 windows -- the
    windows of a few thousand pixels at a time cut on the device (cmlpl_extract_patches) and run through the general
    forward (cmlpl_basenet2_fwd).  Either way the scene stays in HBM as its cube; the 19.9 GB patch tensor never exists."""
    lib = _lib.load()
    cs = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
    return shape.H == shape.W and lib.cmlpl_workspace_bytes(C.byref(cs), 1, 8, 8) > 0


@torch.no_grad()
def infer_cube(net, cube: torch.Tensor, spectra: torch.Tensor, pixel0: int = 0, n: Optional[int] = None,
               chunk: int = 65536, want_logits: bool = False):
    """argmax labels (int64 cuda [n]) of pixels pixel0 .. pixel0 + n - 1 (row-major; default: the whole scene), and the
    logits [n, K] when asked for.  Asynchronous.  Window shapes the fused per-sample forward does not take (more than 256
    window pixels: the reference's 20 x 20) go through ``_infer_cube_by_patches``: same results, ``chunk`` pixels' windows
    in HBM at a time."""
    if not (cube.is_cuda and cube.dtype == torch.float32 and cube.is_contiguous() and cube.dim() == 3):
        raise ValueError("cube: need contiguous float32 cuda tensor [rows, cols, C]")
    rows, cols, Cc = cube.shape
    if not (spectra.is_cuda and spectra.dtype == torch.float32 and spectra.is_contiguous() and spectra.dim() == 2
            and spectra.shape[0] == rows * cols):
        raise ValueError("spectra: need contiguous float32 cuda tensor [rows * cols, bands]")
    cs, flat, packed = _net_buffers(net)
    if Cc != cs.C or spectra.shape[1] != cs.bands:
        raise ValueError(f"cube has {Cc} channels / spectra {spectra.shape[1]} bands, the network wants {cs.C} / {cs.bands}")
    n = rows * cols - pixel0 if n is None else int(n)
    if pixel0 < 0 or n < 1 or pixel0 + n > rows * cols:
        raise ValueError("pixel range outside the scene")
    lib = _lib.load()
    dev = cube.device
    labels = torch.empty(n, dtype=torch.int64, device=dev)
    logits = torch.empty(n, cs.K, dtype=torch.float32, device=dev) if want_logits else None
    chunk = max(8, min(int(chunk), n))
    need = lib.cmlpl_infer_workspace_bytes(C.byref(cs), chunk)
    if need == 0:                                           # not a window the fused per-sample forward takes
        _infer_cube_by_patches(lib, cs, flat, packed, cube, spectra, pixel0, n, min(chunk, 4096), labels, logits)
        return (labels, logits) if want_logits else labels
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        _lib.check("cmlpl_infer_cube", lib.cmlpl_infer_cube(
            C.byref(cs), flat.data_ptr(), packed.data_ptr(), cube.data_ptr(), rows, cols, spectra.data_ptr(),
            pixel0 + o, m, labels.data_ptr() + 8 * o, None if logits is None else logits.data_ptr() + 4 * cs.K * o,
            ws.data_ptr(), ws.numel(), st))
    return (labels, logits) if want_logits else labels


def _infer_cube_by_patches(lib, cs, flat, packed, cube, spectra, pixel0, n, chunk, labels, logits):
    """``chunk`` pixels at a time: their windows cut from the cube on the device (cmlpl_extract_patches: mirror index,
    tools/hyper_tools.py:35-55,226-243), the general eval forward on them (cmlpl_basenet2_fwd, one network, no dropout),
    argmax (first maximum, NaN first: torch.max's rule, as in cmlpl_infer_cube).  One patch buffer of chunk x C x w x w
    floats (393 MB at 4096 pixels of 20 x 20 x 60) is re-used; nothing leaves the device."""
    if cs.H != cs.W:
        raise _lib.CmlplError("cmlpl_infer_cube", -2)
    dev = cube.device
    rows, cols, _ = cube.shape
    need = lib.cmlpl_workspace_bytes(C.byref(cs), 1, chunk, chunk)
    if need == 0:
        raise _lib.CmlplError("cmlpl_workspace_bytes", -2)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    xp = torch.empty(chunk, cs.C, cs.H, cs.W, dtype=torch.float32, device=dev)
    z = torch.empty(chunk, cs.K, dtype=torch.float32, device=dev)
    feat = torch.empty(chunk, 1024, dtype=torch.float32, device=dev)
    idx = torch.arange(pixel0, pixel0 + n, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        _lib.check("cmlpl_extract_patches", lib.cmlpl_extract_patches(
            cube.data_ptr(), rows, cols, cs.C, cs.H, idx.data_ptr() + 8 * o, m, xp.data_ptr(), st))
        _lib.check("cmlpl_basenet2_fwd", lib.cmlpl_basenet2_fwd(
            C.byref(cs), 1, m, flat.data_ptr(), flat.numel(), packed.data_ptr(), xp.data_ptr(),
            spectra.data_ptr() + 4 * cs.bands * (pixel0 + o), None, None, 0.0, 0, 0, 0, None,
            z.data_ptr(), feat.data_ptr(), ws.data_ptr(), ws.numel(), st))
        zz = z[:m]
        nan = torch.isnan(zz)                                   # torch.max: a NaN logit is the maximum, the first one wins
        lab = torch.where(nan.any(1), nan.int().argmax(1), zz.argmax(1))
        labels[o:o + m] = lab
        if logits is not None:
            logits[o:o + m] = zz
