"""On-device patch extraction (reference tools/hyper_tools.py:226-243 ``ExtractPatches``): gathers the
w x w windows of the requested pixels from the scene cube that stays resident in HBM, instead of
materialising and re-reading the [K, C, w, w] patch tensor."""
import ctypes as C

import torch

from . import _lib


def extract_patches(cube: torch.Tensor, pixel_idx: torch.Tensor, w: int, out: torch.Tensor = None) -> torch.Tensor:
    """cube [rows, cols, C] float32 cuda; pixel_idx int64 cuda [n] (row-major pixel numbers) -> [n, C, w, w]."""
    if not (cube.is_cuda and cube.dtype == torch.float32 and cube.is_contiguous() and cube.dim() == 3):
        raise ValueError("cube: need contiguous float32 cuda tensor [rows, cols, C]")
    if not (pixel_idx.is_cuda and pixel_idx.dtype == torch.int64 and pixel_idx.is_contiguous()):
        raise ValueError("pixel_idx: need contiguous int64 cuda tensor")
    rows, cols, Cc = cube.shape
    n = pixel_idx.numel()
    if out is None:
        out = torch.empty(n, Cc, w, w, device=cube.device, dtype=torch.float32)
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream(cube.device).cuda_stream)
    _lib.check("cmlpl_extract_patches", lib.cmlpl_extract_patches(
        cube.data_ptr(), rows, cols, Cc, int(w), pixel_idx.data_ptr(), n, out.data_ptr(), st))
    return out
