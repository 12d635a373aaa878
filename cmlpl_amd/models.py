"""Drop-in ``BaseNet2`` / ``Normalize`` (reference tools/models.py:81-90,97-152) on the HIP path.

Same constructor, same 16 ``state_dict`` keys and shapes, same
``forward(x, y) -> (logits, l2-normalised 1024-d feature)``; forward and backward run in
libcmlpl_hip.so (``cmlpl_basenet2_fwd`` / ``_bwd``).  Extra keyword arguments generalise the two
literals the reference hard-codes (60 input channels, 2624-wide classifier) so the BASELINE
window shapes can be built; defaults reproduce the reference exactly.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from .config import FEAT_DIM, NetShape


class Normalize(nn.Module):
    """x / ||x||_p along dim 1, no epsilon (reference tools/models.py:81-90)."""

    def __init__(self, power=2):
        super().__init__()
        self.power = power

    def forward(self, x):
        return x.div(x.pow(self.power).sum(1, keepdim=True).pow(1.0 / self.power))


class _BaseNet2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, y, dropmask, *params):
        lib = _lib.load()
        n = x.shape[0]
        flat, packed = mod._flat_params(params)
        ws = mod._workspace(n)
        K = mod.shape.K
        logits = torch.empty(1, n, K, device=x.device, dtype=torch.float32)
        feat = torch.empty(1, n, FEAT_DIM, device=x.device, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        train = 1 if mod.training else 0
        mod._calls += 1
        _lib.check("cmlpl_basenet2_fwd", lib.cmlpl_basenet2_fwd(
            C.byref(mod._cshape), 1, n, flat.data_ptr(), mod._P, packed.data_ptr(), x.data_ptr(), y.data_ptr(), None,
            None if dropmask is None else dropmask.data_ptr(), float(mod.dropout), train,
            int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, mod._calls, None,
            logits.data_ptr(), feat.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        ctx.mod, ctx.n, ctx.train = mod, n, train
        ctx.save_for_backward(x, y, dropmask if dropmask is not None else torch.empty(0), flat, packed, ws)
        return logits[0], feat[0]

    @staticmethod
    def backward(ctx, dlogits, dfeat):
        lib = _lib.load()
        mod, n = ctx.mod, ctx.n
        x, y, dropmask, flat, packed, ws = ctx.saved_tensors
        dlogits = dlogits.contiguous() if dlogits is not None else torch.zeros(n, mod.shape.K, device=x.device)
        dfeat_ptr = None
        if dfeat is not None:
            dfeat = dfeat.contiguous()
            dfeat_ptr = dfeat.data_ptr()
        grads = torch.zeros(mod._P, device=x.device, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check("cmlpl_basenet2_bwd", lib.cmlpl_basenet2_bwd(
            C.byref(mod._cshape), 1, n, flat.data_ptr(), mod._P, packed.data_ptr(), x.data_ptr(), y.data_ptr(),
            dropmask.data_ptr() if dropmask.numel() else None, float(mod.dropout), ctx.train,
            dlogits.data_ptr(), dfeat_ptr, grads.data_ptr(), mod._P, ws.data_ptr(), ws.numel(), stream))
        out = []
        for i in range(_lib.NUM_LIVE):
            off, numel = int(mod._layout.param_off[i]), int(mod._layout.param_numel[i])
            out.append(grads[off:off + numel].view(mod._live_shapes[i]))
        return (None, None, None, None, *out)


class BaseNet2(nn.Module):
    """reference tools/models.py:97-152.  ``in_channels`` / ``window`` default to the reference
    literals (60, 20); other values build the generalised net of SURVEY.md section 0."""

    def __init__(self, num_features=103, dropout=0, num_classes=0, in_channels=60, window=20):
        super().__init__()
        H = W = int(window)
        self.shape = NetShape(int(in_channels), H, W, int(num_features), int(num_classes))
        self.num_features, self.dropout, self.num_classes = num_features, dropout, num_classes
        # parameter containers only (never called): same names/shapes/default init as the reference
        self.conv0 = nn.Conv2d(in_channels, 64, kernel_size=1, stride=1, bias=True)
        self.conv1 = nn.Conv2d(64, 64, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=3, stride=1, padding=1, bias=True)
        self.feat_spe = nn.Linear(num_features, FEAT_DIM)
        self.feat_ss = nn.Linear(FEAT_DIM, 256)      # constructed but unused in forward, like the
        self.feat_ss2 = nn.Linear(FEAT_DIM, 64)      # reference (models.py:122-126): they stay in
        self.feat_ss3 = nn.Linear(256, 64)           # state_dict() and never receive a gradient
        self.classifier = nn.Linear(self.shape.cls_in, num_classes)
        self.l2norm = Normalize(2)
        self._cshape = _lib.Shape(self.shape.C, H, W, self.shape.bands, self.shape.K)
        self._layout = None
        self._cache = None
        self._calls = 0

    def _live_params(self):
        return (self.conv0.weight, self.conv0.bias, self.conv1.weight, self.conv1.bias, self.conv2.weight,
                self.conv2.bias, self.feat_spe.weight, self.feat_spe.bias, self.classifier.weight,
                self.classifier.bias)

    def _flat_params(self, params):
        """Flat [P] copy of the live tensors + packed conv weights, rebuilt only when a parameter changed."""
        lib = _lib.load()
        if self._layout is None:
            self._layout = _lib.layout(self._cshape)
            self._P = int(self._layout.param_total)
            self._live_shapes = [tuple(p.shape) for p in params]
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._cache is None or self._cache[0] != key:
            dev = params[0].device
            flat = torch.zeros(self._P, device=dev, dtype=torch.float32)
            for i, p in enumerate(params):
                off, numel = int(self._layout.param_off[i]), int(self._layout.param_numel[i])
                flat[off:off + numel].copy_(p.detach().reshape(-1))
            packed = torch.empty(int(self._layout.packed_total), device=dev, dtype=torch.float32)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check("cmlpl_pack_weights", lib.cmlpl_pack_weights(C.byref(self._cshape), 1, flat.data_ptr(),
                                                                     self._P, packed.data_ptr(), stream))
            self._cache = (key, flat, packed)
        return self._cache[1], self._cache[2]

    def _workspace(self, n):
        lib = _lib.load()
        need = lib.cmlpl_workspace_bytes(C.byref(self._cshape), 1, n, n)
        if need == 0:
            raise _lib.CmlplError("cmlpl_workspace_bytes", -2)
        dev = self.conv0.weight.device
        # one workspace per forward call, kept alive by the autograd context that saved it: any number of
        # forwards may run before their backwards (gradient accumulation, an eval pass in between), like nn.Module
        return torch.empty(need, dtype=torch.uint8, device=dev)

    def forward(self, x, y, dropmask=None):
        """x: [n, C, H, W] patch windows, y: [n, bands] spectra -> (logits [n,K], feat [n,1024]).
        ``dropmask`` (optional, [n, cls_in]) injects an explicit dropout multiplier (parity tests)."""
        if not x.is_cuda:
            raise RuntimeError("cmlpl_amd.BaseNet2 runs on the GPU only (no CPU fallback); call .cuda()")
        s = self.shape
        if tuple(x.shape[1:]) != (s.C, s.H, s.W) or tuple(y.shape[1:]) != (s.bands,):
            raise ValueError(f"expected x [n,{s.C},{s.H},{s.W}] and y [n,{s.bands}], got {tuple(x.shape)} {tuple(y.shape)}")
        x = x.contiguous().float()
        y = y.contiguous().float()
        if dropmask is not None:
            dropmask = dropmask.contiguous().float()
        return _BaseNet2Fn.apply(self, x, y, dropmask, *self._live_params())


class _NTXentFn(torch.autograd.Function):
    """One C call forms the loss AND both gradients.  Host cost matters here (two launches of 13 us each at B = 128): the
    workspace is kept per (B, D, device), the two gradients share one allocation, and the backward scales them with one
    launch instead of two."""
    _ws = {}

    @staticmethod
    def forward(ctx, emb_i, emb_j, temperature):
        lib = _lib.load()
        B, D = emb_i.shape
        dev = emb_i.device
        key = (B, D, dev)
        ws = _NTXentFn._ws.get(key)
        if ws is None:     # (stream-ordered re-use: a later call's kernels run behind this one's on the same stream)
            ws = _NTXentFn._ws[key] = torch.empty(lib.cmlpl_ntxent_workspace_bytes(B, D), dtype=torch.uint8, device=dev)
        out = torch.empty(2 * B * D + 1, device=dev, dtype=torch.float32)        # [gi | gj | loss]
        g2 = out[:2 * B * D].view(2, B, D)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = out.data_ptr()
        _lib.check("cmlpl_ntxent_fwd_bwd", lib.cmlpl_ntxent_fwd_bwd(
            emb_i.data_ptr(), emb_j.data_ptr(), B, D, temperature, p + 8 * B * D, p, p + 4 * B * D,
            ws.data_ptr(), ws.numel(), stream))
        ctx.g2 = g2
        return out[2 * B * D]

    @staticmethod
    def backward(ctx, g):
        s = ctx.g2 * g
        return s[0], s[1], None


class ContrastiveLoss(nn.Module):
    """Drop-in for the reference's tools.models.ContrastiveLoss (tools/models.py:14-39): same constructor
    ``(batch_size, device='cuda', temperature=0.5)`` and ``forward(emb_i, emb_j) -> scalar``; forward and
    backward run in libcmlpl_hip.so (cmlpl_ntxent_fwd_bwd)."""

    def __init__(self, batch_size, device="cuda", temperature=0.5):
        super().__init__()
        self.batch_size = batch_size
        self._temperature = float(temperature)       # host copy: float(buffer) would synchronise with the device per call
        self.register_buffer("temperature", torch.tensor(float(temperature)))
        # state_dict() parity with the reference (models.py:19-20); the kernel masks the diagonal itself
        self.register_buffer("negatives_mask", (~torch.eye(batch_size * 2, batch_size * 2, dtype=torch.bool)).float())

    def forward(self, emb_i, emb_j):
        if not emb_i.is_cuda:
            raise RuntimeError("cmlpl_amd.ContrastiveLoss runs on the GPU only (no CPU fallback)")
        if emb_i.shape != emb_j.shape or emb_i.shape[0] != self.batch_size:
            raise ValueError("emb_i / emb_j must both be [batch_size, D]")
        return _NTXentFn.apply(emb_i.contiguous().float(), emb_j.contiguous().float(), self._temperature)
