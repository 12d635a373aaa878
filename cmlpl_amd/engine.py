"""TrainEngine: the CMLPL training step (reference train.py:150-278) on one MI355X.

Host logic only -- state ownership (flat parameter / Adam / bank buffers as torch
tensors), the bank-pointer bookkeeping of train.py:234,237, the schedule scalars of
train.py:147-148,212 -- and ONE ctypes call per step into ``cmlpl_train_step``
(libcmlpl_hip.so).  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import torch

import os
import weakref

from . import _lib
from .config import FEAT_DIM, HyperParams, NetShape

_CHECK_INDICES = os.environ.get("CMLPL_CHECK_INDICES", "0") not in ("", "0")

SCALAR_NAMES = ("ctr_s", "total_s", "cls_s", "con_s", "acc", "total_w", "cls_w", "con_w", "ctr_w",
                "n_mask_w", "n_mask_s", "n_pos", "n_neg")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk_f32(t: torch.Tensor, shape, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != tuple(shape) or not t.is_cuda:
        raise ValueError(f"{name}: need contiguous float32 cuda tensor of shape {tuple(shape)}, "
                         f"got {t.dtype} {tuple(t.shape)} on {t.device}")


class TrainEngine:
    """Both networks (Base = net 0 / "s", Base1 = net 1 / "w"), their Adam state and
    the two memory banks, resident in HBM for the whole run.

    ``labeled_batch_size`` sizes the banks exactly like train.py:138
    (``queue_size = 5 * labeled_batch_size * 2``).  Bank writes wrap modulo the bank
    size where the reference's slice-assign would raise; the pointer advance is the
    reference literal 256 (``hp.bank_step``) -- see DESIGN.md "memory bank".
    """
    takes_indices = True      # step(..., lab_idx=, unl_idx=): batches as row indices into the resident splits

    def __init__(self, shape: NetShape, labeled_batch_size: int, unlabeled_batch_size: int,
                 hp: Optional[HyperParams] = None, device="cuda:0", seed: int = 1088, bank_labeled: int = 0,
                 hist_rows: int = 1):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("cmlpl_amd.TrainEngine needs a GPU (no CPU fallback)")
        self.shape, self.hp = shape, hp or HyperParams()
        self.device = torch.device(device)
        self.bt_max, self.btu_max = int(labeled_batch_size), int(unlabeled_batch_size)
        self.n_max = self.bt_max + self.btu_max
        self.cshape = _lib.Shape(shape.C, shape.H, shape.W, shape.bands, shape.K)
        self.layout = _lib.layout(self.cshape)
        L = self.layout
        self.P = int(L.param_total)
        # train.py:138 (bank_labeled: the GLOBAL labelled batch under data parallelism)
        self.Q = self.hp.bank_mult * (bank_labeled or self.bt_max) * 2
        if self.Q < self.n_max:
            raise ValueError("bank smaller than one batch (needs 10*bt >= bt+btu)")
        dev = self.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.params, self.m, self.v, self.grads = z(2, self.P), z(2, self.P), z(2, self.P), z(2, self.P)
        self.packed = z(2, int(L.packed_total))
        self.bank_feats, self.bank_probs = z(2, self.Q, FEAT_DIM), z(2, self.Q, shape.K)   # train.py:139-144
        self.ptr = [0, 0]                                                   # train.py:141,145
        # Scalars of step i land in row i % hist_rows of a device-side ring (loss_hist of train.py:136,274-278 without
        # the reference's five .item() syncs per step): the step writes its row directly, the host reads a window
        # back once per print_per_batches steps.
        self.hist_rows = max(1, int(hist_rows))
        self.scalar_hist = z(self.hist_rows, 16)
        self._cur_row = 0
        self.logits, self.feat = z(2, self.n_max, shape.K), z(2, self.n_max, FEAT_DIM)
        ws = self.lib.cmlpl_workspace_bytes(C.byref(self.cshape), 2, self.n_max, self.Q)
        if ws == 0:
            raise _lib.CmlplError("cmlpl_workspace_bytes", -2)
        self.workspace = torch.empty(ws, dtype=torch.uint8, device=dev)
        self.adam_t = 0
        self.step_count = 0
        self.seed = int(seed)
        self._chp = self._make_hp()
        self._packed_dirty = True
        # the step's argument record is kept: what never changes (state buffers, sizes) is set once, a step writes the
        # rest -- building the record anew and re-checking the same resident tensors cost ~40 us of Python per step
        self._io = _lib.StepIO()
        self._fill_state(self._io)
        self._checked = None
        self._graph = None          # weak reference to the StepGraph that has programmed replays (see step())

    @property
    def scalars(self) -> torch.Tensor:
        """the logged row of the last step (a view of the device ring)"""
        return self.scalar_hist[self._cur_row]

    # ------------------------------------------------------------------ parameters
    def _make_hp(self) -> _lib.HParams:
        h = self.hp
        return _lib.HParams(h.lr, h.beta1, h.beta2, h.eps, h.temperature, h.alpha, h.noise, h.dropout,
                            h.w_contrast, h.w_mutual, h.pos_thr, h.neg_thr)

    def tensor_shapes(self) -> "OrderedDict[str, tuple]":
        s = self.shape
        return OrderedDict([
            ("conv0.weight", (64, s.C, 1, 1)), ("conv0.bias", (64,)),
            ("conv1.weight", (64, 64, 3, 3)), ("conv1.bias", (64,)),
            ("conv2.weight", (64, 64, 3, 3)), ("conv2.bias", (64,)),
            ("feat_spe.weight", (FEAT_DIM, s.bands)), ("feat_spe.bias", (FEAT_DIM,)),
            ("classifier.weight", (s.K, s.cls_in)), ("classifier.bias", (s.K,)),
            ("feat_ss.weight", (256, FEAT_DIM)), ("feat_ss.bias", (256,)),
            ("feat_ss2.weight", (64, FEAT_DIM)), ("feat_ss2.bias", (64,)),
            ("feat_ss3.weight", (64, 256)), ("feat_ss3.bias", (64,)),
        ])

    def view(self, buf: torch.Tensor, net: int, key: str) -> torch.Tensor:
        i = _lib.TENSOR_KEYS.index(key)
        off, numel = int(self.layout.param_off[i]), int(self.layout.param_numel[i])
        return buf[net, off:off + numel].view(self.tensor_shapes()[key])

    def load_state_dict(self, net: int, sd: Dict[str, torch.Tensor]) -> None:
        for key in _lib.TENSOR_KEYS:
            if key in sd:
                self.view(self.params, net, key).copy_(sd[key].to(self.device, torch.float32))
        self._packed_dirty = True

    def state_dict(self, net: int) -> "OrderedDict[str, torch.Tensor]":
        order = ["conv0", "conv1", "conv2", "feat_spe", "feat_ss", "feat_ss2", "feat_ss3", "classifier"]
        out = OrderedDict()
        for mod in order:                      # registration order of tools/models.py:102-127
            for leaf in ("weight", "bias"):
                out[f"{mod}.{leaf}"] = self.view(self.params, net, f"{mod}.{leaf}").detach().clone()
        return out

    def grad(self, net: int, key: str) -> torch.Tensor:
        return self.view(self.grads, net, key)

    def init_params_default(self, seed: int = 1088) -> None:
        """torch default init of nn.Conv2d / nn.Linear (kaiming-uniform a=sqrt(5)):
        U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        shapes = self.tensor_shapes()
        for net in range(2):
            for key, shp in shapes.items():
                fan_in = 1
                for d in shapes[key.split(".")[0] + ".weight"][1:]:
                    fan_in *= d
                b = 1.0 / (fan_in ** 0.5)
                self.view(self.params, net, key).copy_((torch.rand(shp, generator=g) * 2 - 1) * b)
        self._packed_dirty = True

    def _ensure_packed(self, stream) -> None:
        if self._packed_dirty:
            _lib.check("cmlpl_pack_weights",
                       self.lib.cmlpl_pack_weights(C.byref(self.cshape), 2, _ptr(self.params), self.P,
                                                   _ptr(self.packed), stream))
            self._packed_dirty = False

    # ------------------------------------------------------------------ the step
    def _check_rows(self, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx):
        """Shapes of a batch: the rows themselves, or (lab_idx / unl_idx given) the resident splits the int64 index
        lists point into (hsi_loader.py:109-133 hands rows out by index; here the kernels follow the index)."""
        s = self.shape
        if lab_idx is not None and unl_idx is not None:
            # the same resident splits as last time (by identity and storage): only the index lists are new
            key = (id(XPl), id(Xl), id(Y), id(XPu), id(Xu), XPl.data_ptr(), XPu.data_ptr(), XPl.shape[0], XPu.shape[0])
            if key == self._checked and lab_idx.dtype == torch.int64 and unl_idx.dtype == torch.int64 \
                    and lab_idx.is_cuda and unl_idx.is_cuda and lab_idx.dim() == 1 and unl_idx.dim() == 1 \
                    and lab_idx.is_contiguous() and unl_idx.is_contiguous():
                bt, btu = lab_idx.shape[0], unl_idx.shape[0]
                if 1 <= bt <= self.bt_max and btu >= 1 and bt + btu <= self.n_max:
                    return bt, btu
        if (lab_idx is None) != (unl_idx is None):
            raise ValueError("lab_idx and unl_idx come together")
        for name, t in (("lab_idx", lab_idx), ("unl_idx", unl_idx)):
            if t is not None and (t.dtype != torch.int64 or t.dim() != 1 or not t.is_cuda or not t.is_contiguous()):
                raise ValueError(f"{name}: need a contiguous int64 cuda vector")
        bt = XPl.shape[0] if lab_idx is None else lab_idx.shape[0]
        btu = XPu.shape[0] if unl_idx is None else unl_idx.shape[0]
        n = bt + btu
        if bt < 1 or btu < 1 or bt > self.bt_max or n > self.n_max:
            raise ValueError(f"batch {bt}+{btu} outside the engine's capacity {self.bt_max}+{self.btu_max}")
        nl, nu = XPl.shape[0], XPu.shape[0]
        _chk_f32(XPl, (nl, s.C, s.H, s.W), "XPl"); _chk_f32(Xl, (nl, s.bands), "Xl")
        _chk_f32(XPu, (nu, s.C, s.H, s.W), "XPu"); _chk_f32(Xu, (nu, s.bands), "Xu")
        if Y.dtype != torch.int64 or tuple(Y.shape) != (nl,) or not Y.is_cuda:
            raise ValueError("Y: need int64 cuda tensor, one label per labelled row")
        if lab_idx is not None:
            self._checked = (id(XPl), id(Xl), id(Y), id(XPu), id(Xu), XPl.data_ptr(), XPu.data_ptr(), XPl.shape[0], XPu.shape[0])
        return bt, btu

    @staticmethod
    def check_index_range(idx: torch.Tensor, rows: int, name: str = "index") -> None:
        """Every entry of an index list inside [0, rows): the kernels follow the indices without a bounds check, so a
        stale or corrupt permutation would be an out-of-bounds device read.  One synchronising min / max: call it
        where a list is FILLED (the loader's per-epoch permutation, StepGraph's buffers), not per step;
        ``CMLPL_CHECK_INDICES=1`` makes ``step()`` call it on every batch (bring-up only)."""
        if idx.numel() == 0:
            return
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= rows:
            raise ValueError(f"{name}: entries span [{lo}, {hi}], the resident split has {rows} rows")

    def _fill_io(self, io, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt, btu):
        io.d_xpl, io.d_xl, io.d_labels = XPl.data_ptr(), Xl.data_ptr(), Y.data_ptr()
        io.d_xpu, io.d_xu = XPu.data_ptr(), Xu.data_ptr()
        io.d_lab_idx = None if lab_idx is None else lab_idx.data_ptr()
        io.d_unl_idx = None if unl_idx is None else unl_idx.data_ptr()
        io.bt, io.btu = bt, btu

    def _fill_state(self, io):
        io.d_params, io.d_m, io.d_v = self.params.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
        io.d_grads, io.d_packed = self.grads.data_ptr(), self.packed.data_ptr()
        for i in range(2):
            io.banks.d_feats[i] = self.bank_feats[i].data_ptr()
            io.banks.d_probs[i] = self.bank_probs[i].data_ptr()
        io.banks.Q = self.Q
        # outputs are laid out [2][n][..] for THIS n (views of the max-size buffers)
        io.d_logits, io.d_feat = self.logits.data_ptr(), self.feat.data_ptr()
        io.d_workspace, io.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel()
        io.seed = self.seed

    def _advance(self, n, apply_update=True):
        """host copy of the step bookkeeping: bank pointers (train.py:234,237 -- ptr1 follows ptr0, reference quirk
        kept), Adam step, step counter"""
        p0 = (self.ptr[0] + self.hp.bank_step) % self.Q
        self.ptr = [p0, (p0 + self.hp.bank_step) % self.Q]
        if apply_update:
            self.adam_t += 1
        self.step_count += 1
        self._last_n = n

    def step(self, XPl: torch.Tensor, Xl: torch.Tensor, Y: torch.Tensor, XPu: torch.Tensor, Xu: torch.Tensor,
             epoch: int, batch_index: int, noise: Optional[Sequence[torch.Tensor]] = None,
             dropmask: Optional[torch.Tensor] = None, apply_update: bool = True,
             lab_idx: Optional[torch.Tensor] = None, unl_idx: Optional[torch.Tensor] = None) -> None:
        """One training step; asynchronous.  Results land in ``self.scalars`` (device),
        ``self.logits`` / ``self.feat`` ([2][n][..]) and ``self.grads``.

        noise    : None -> in-kernel draws (PCG4D hash + Box-Muller); or the 8 draws in reference order (parity mode)
        dropmask : None -> Philox4x32-10 mask (or no dropout when hp.dropout == 0); or [2][n][cls_in] multipliers
        lab_idx / unl_idx : None -> XPl .. Xu ARE the batch; or int64 row numbers: XPl / Xl / Y (XPu / Xu) are then the
                   whole resident labelled (unlabelled) split and batch row s is its row lab_idx[s] (unl_idx[s]) --
                   no gathered copy of the batch is made (noise / dropmask stay indexed by batch row)
        """
        s = self.shape
        g = self._graph() if self._graph is not None else None
        if g is not None and g.pending > 0:
            # the remaining replays were programmed from the state BEFORE this step (counters, bank pointers, Adam step)
            raise RuntimeError(f"{g.pending} programmed graph replays are pending: launch them (or program() anew) "
                               "before an eager step")
        bt, btu = self._check_rows(XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx)
        if _CHECK_INDICES and lab_idx is not None:
            self.check_index_range(lab_idx, XPl.shape[0], "lab_idx")
            self.check_index_range(unl_idx, XPu.shape[0], "unl_idx")
        n = bt + btu
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._ensure_packed(stream)
        io = self._io
        self._fill_io(io, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt, btu)
        io.noise8, io.d_dropmask = None, None
        keep = None
        if noise is not None:
            xpl, xl, xpu, xu = (bt, s.C, s.H, s.W), (bt, s.bands), (btu, s.C, s.H, s.W), (btu, s.bands)
            shp = [xpl, xl, xpl, xl, xpu, xu, xpu, xu]
            for t, sh in zip(noise, shp):
                _chk_f32(t, sh, "noise")
            keep = (C.c_void_p * 8)(*[t.data_ptr() for t in noise])
            io.noise8 = C.cast(keep, C.POINTER(C.c_void_p))
        if dropmask is not None:
            _chk_f32(dropmask, (2, n, s.cls_in), "dropmask")
            io.d_dropmask = dropmask.data_ptr()
        for i in range(2):
            io.banks.ptr[i] = self.ptr[i]
        self._cur_row = self.step_count % self.hist_rows
        io.d_scalars = self.scalar_hist.data_ptr() + 64 * self._cur_row
        io.smooth = 1 if self.hp.smooth_gate(epoch, batch_index) else 0
        io.adap_mask = float(self.hp.thr * self.hp.adap_thr(epoch))        # train.py:221
        io.adam_t = self.adam_t + 1
        io.step = self.step_count
        io.apply_update = 1 if apply_update else 0
        _lib.check("cmlpl_train_step",
                   self.lib.cmlpl_train_step(C.byref(self.cshape), C.byref(self._chp), C.byref(io), stream))
        self._advance(n, apply_update)

    def capture(self, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt: int, btu: int, capacity: int = 1024) -> "StepGraph":
        """The step captured ONCE as a hipGraph over the resident splits and two index buffers (lab_idx / unl_idx: the
        epoch's permutations, re-filled in place by the caller); see StepGraph."""
        return StepGraph(self, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt, btu, capacity)

    def outputs(self):
        """(logits [2][n][K], feat [2][n][1024]) of the last step."""
        n = self._last_n
        K = self.shape.K
        lo = self.logits.view(-1)[: 2 * n * K].view(2, n, K)
        fe = self.feat.view(-1)[: 2 * n * FEAT_DIM].view(2, n, FEAT_DIM)
        return lo, fe

    def debug_region(self, name: str, dtype=torch.float32) -> torch.Tensor:
        """View of a saved activation of the last step inside the workspace (inspection / tests);
        see cmlpl_debug_region in include/cmlpl.h."""
        off, nbytes = C.c_size_t(), C.c_size_t()
        _lib.check("cmlpl_debug_region", self.lib.cmlpl_debug_region(
            C.byref(self.cshape), 2, self._last_n, name.encode(), C.byref(off), C.byref(nbytes)))
        return self.workspace[off.value: off.value + nbytes.value].view(dtype)

    def read_scalars(self) -> Dict[str, float]:
        """Synchronising read of the logged row (train.py:274-278) and friends."""
        vals = self.scalars.tolist()
        return dict(zip(SCALAR_NAMES, vals))

    def loss_row(self):
        """[loss_contrast, total_loss, cls_loss, con_loss, acc] -- loss_hist row, train.py:274-278."""
        return self.scalars[:5].tolist()

    def loss_window(self, k: int):
        """loss_hist[index_i-k+1 : index_i+1] (train.py:285-289) as a float64 numpy array [k,5]: the rows of the
        last k steps, oldest first.  One synchronising read-back."""
        if k < 1 or k > self.hist_rows or k > self.step_count:
            raise ValueError(f"window of {k} steps not held (hist_rows={self.hist_rows}, steps={self.step_count})")
        idx = [(self.step_count - k + j) % self.hist_rows for j in range(k)]
        rows = self.scalar_hist[torch.tensor(idx, device=self.device)]
        return self._reduce_rows(rows)[:, :5].double().cpu().numpy()

    def _reduce_rows(self, rows: torch.Tensor) -> torch.Tensor:
        return rows


class StepGraph:
    """The training step as a replayable hipGraph (SURVEY.md section 7 stage 6; ``cmlpl_step_graph_create``).

    Everything that changes from step to step -- random-stream counter, Adam step, bank pointers, the train.py:212 /
    :221 gates, the offsets of the batch inside the index buffers, the row of the logging ring -- lives in a device
    table of ``cmlpl_dyn`` rows that ``program()`` fills for a run of steps ahead of time (an epoch, say); a device
    cursor walks it, advanced by the step itself.  ``launch()`` is then ONE hipGraphLaunch per step: no arguments to
    marshal, nothing else enqueued.  The engine's host-side bookkeeping advances exactly as in ``TrainEngine.step``,
    so eager steps and replays can be mixed (an epoch's short last batch runs eagerly: its shape is not the graph's)
    -- but only BETWEEN programs: an eager step while programmed replays are pending raises (their rows were formed
    from the state before it).
    """

    def __init__(self, eng: TrainEngine, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt: int, btu: int, capacity: int = 1024):
        import numpy as np
        self.eng, self.bt, self.btu, self.capacity = eng, int(bt), int(btu), int(capacity)
        if lab_idx is None or unl_idx is None:
            raise ValueError("a captured step reads its rows through index buffers")
        if lab_idx.shape[0] < bt or unl_idx.shape[0] < btu:
            raise ValueError("index buffers shorter than one batch")
        for name, t in (("lab_idx", lab_idx), ("unl_idx", unl_idx)):
            if t.dtype != torch.int64 or t.dim() != 1 or not t.is_cuda or not t.is_contiguous():
                raise ValueError(f"{name}: need a contiguous int64 cuda vector")
        eng._check_rows(XPl, Xl, Y, XPu, Xu, lab_idx[:bt], unl_idx[:btu])
        self._keep = (XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx)
        self.validate_indices()
        self.n_lab_idx, self.n_unl_idx = int(lab_idx.shape[0]), int(unl_idx.shape[0])
        dev = eng.device
        # row 0 = the working copy the kernels read, rows 1 .. k the programmed steps, one spare row behind them
        self.table = torch.zeros((self.capacity + 2) * 64, dtype=torch.uint8, device=dev)
        self.cursor = torch.ones(1, dtype=torch.int32, device=dev)
        self.host = torch.zeros((self.capacity + 2) * 64, dtype=torch.uint8).pin_memory()
        self.rows = self.host.numpy().view(np.dtype(_lib.DYN_DTYPE))[1:]
        self.pending = 0
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        eng._ensure_packed(stream)
        io = _lib.StepIO()
        eng._fill_state(io)
        eng._fill_io(io, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, self.bt, self.btu)
        io.d_scalars = eng.scalar_hist.data_ptr()          # ring base: the row comes from the table
        io.apply_update = 1
        io.d_dyn_table, io.d_dyn_cursor = self.table.data_ptr(), self.cursor.data_ptr()
        self._io = io
        # the launchers set their kernels' LDS attributes on first use, which must not happen inside a capture
        if eng.step_count == 0:
            raise RuntimeError("run one eager TrainEngine.step() before capturing (kernel attributes are set lazily)")
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        handle = C.c_void_p()
        with torch.cuda.stream(cap):
            _lib.check("cmlpl_step_graph_create", eng.lib.cmlpl_step_graph_create(
                C.byref(eng.cshape), C.byref(eng._chp), C.byref(io), C.c_void_p(cap.cuda_stream), C.byref(handle)))
        torch.cuda.current_stream(dev).wait_stream(cap)
        self.handle = handle

    def validate_indices(self) -> None:
        """Range check of the two index buffers (synchronising): at capture, and whenever the caller has re-filled them
        in place and wants the check (the kernels follow the indices blindly)."""
        XPl, _, _, XPu, _, lab_idx, unl_idx = self._keep
        TrainEngine.check_index_range(lab_idx, XPl.shape[0], "lab_idx")
        TrainEngine.check_index_range(unl_idx, XPu.shape[0], "unl_idx")

    def program(self, steps) -> None:
        """``steps``: (epoch, batch_index, lab_off, unl_off) of the next replays, in order.  Fills the table from the
        engine's current state and rewinds the cursor (stream-ordered: behind every replay enqueued so far)."""
        eng, hp = self.eng, self.eng.hp
        steps = list(steps)
        if self.pending:
            raise RuntimeError(f"{self.pending} programmed steps have not been launched")
        if not 1 <= len(steps) <= self.capacity:
            raise ValueError(f"1..{self.capacity} steps per program")
        if getattr(self, "_copied", None) is not None:
            self._copied.synchronize()        # the previous program's staging rows have left the pinned buffer
        import numpy as np
        k = len(steps)
        st = np.asarray(steps, dtype=np.int64).reshape(k, 4)
        if (st[:, 2:] < 0).any() or (st[:, 2] + self.bt > self.n_lab_idx).any() or (st[:, 3] + self.btu > self.n_unl_idx).any():
            raise ValueError("batch offsets outside the index buffers")
        j = np.arange(k, dtype=np.int64)
        rows = self.rows[:k]
        rows["step"] = eng.step_count + j
        rows["adam_t"] = eng.adam_t + 1 + j
        rows["lab_off"], rows["unl_off"] = st[:, 2], st[:, 3]
        # bank pointers BEFORE step j (train.py:234,237: ptr0 advances by the literal step, ptr1 follows ptr0)
        p0 = (eng.ptr[0] + j * hp.bank_step) % eng.Q
        rows["ptr"][:, 0] = p0
        rows["ptr"][:, 1] = (p0 + hp.bank_step) % eng.Q
        rows["ptr"][0, 1] = eng.ptr[1]
        rows["smooth"] = [1 if hp.smooth_gate(int(e), int(bi)) else 0 for e, bi in st[:, :2]]
        adap = {int(e): float(hp.thr * hp.adap_thr(int(e))) for e in np.unique(st[:, 0])}
        rows["adap_mask"] = [adap[int(e)] for e in st[:, 0]]
        rows["hist_row"] = (eng.step_count + j) % eng.hist_rows
        a, b = C.c_float(), C.c_float()       # Adam's two scalars: formed by the library, exactly as the eager step does
        ss, bc = np.empty(k, np.float32), np.empty(k, np.float32)
        for i in range(k):
            eng.lib.cmlpl_dyn_adam(C.byref(eng._chp), eng.adam_t + 1 + i, C.byref(a), C.byref(b))
            ss[i], bc[i] = a.value, b.value
        rows["adam_step_size"], rows["adam_bc2_sqrt"] = ss, bc
        self.host[:64].copy_(self.host[64:128])            # row 0: the working copy starts as the first step's row
        k = (k + 1) * 64
        self.table[:k].copy_(self.host[:k], non_blocking=True)
        self.cursor.fill_(1)
        self.pending = len(steps)
        eng._graph = weakref.ref(self)
        # (the pinned staging rows may be rewritten only after that copy has run)
        self._copied = torch.cuda.Event()
        self._copied.record(torch.cuda.current_stream(eng.device))

    def launch(self) -> None:
        """one replay = one training step (asynchronous)"""
        eng = self.eng
        if self.pending < 1:
            raise RuntimeError("no programmed step left: call program() first")
        stream = C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
        eng._ensure_packed(stream)            # set_params / load_state_dict between replays: the packed copies follow
        _lib.check("cmlpl_step_graph_launch", eng.lib.cmlpl_step_graph_launch(self.handle, stream))
        eng._cur_row = eng.step_count % eng.hist_rows
        eng._advance(self.bt + self.btu, True)
        self.pending -= 1

    def close(self) -> None:
        if self.handle:
            self.eng.lib.cmlpl_step_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
