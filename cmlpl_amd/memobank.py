"""Host side of the reference's loss_helper.py over the C ABI (include/cmlpl.h).

`MemoryBank` keeps the per-class FIFOs of compute_contra_memobank_loss resident in HBM as rings; the functions
below have the reference's semantics (loss_helper.py:19-36, 39-219, 242-261), data-dependent control flow
(which classes are valid, how many keys were enqueued) is read back from the device exactly where the reference
calls .item().  torch is used for allocation, the random index draws and trivial elementwise glue only; all
selection / gather / similarity / loss arithmetic runs in the HIP kernels of csrc/memobank.hip."""
import ctypes as C

import torch

from . import _lib

NUM_QUERIES = 256          # loss_helper.py:60
NUM_NEGATIVES = 50         # :61
TEMP = 0.5                 # :59
PENDING_MAX = 64           # one-pass calls whose enqueue logs may wait on the device before they are folded in


def _draw_key():
    """64-bit key of one call's in-kernel index draws (the reference's two torch.randint calls, :164/:179).  It comes
    from torch's CPU generator, so it ADVANCES with every call in the process -- across MemoryBank objects (the
    drop-in list path builds a new one per call), across resumed runs that restore the generator state -- and
    torch.manual_seed() still makes a run reproducible.  No device sync."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class MemoryBank:
    """One FIFO of at most queue_size[c] feature rows per class (the reference's `memobank[c][0]`), stored as a
    ring: logical row i of class c is slot (head[c] + i) % queue_size[c]."""

    def __init__(self, num_classes, queue_size, dim, device="cuda"):
        sizes = [int(queue_size)] * num_classes if isinstance(queue_size, int) else [int(s) for s in queue_size]
        assert len(sizes) == num_classes and min(sizes) >= 1
        self.K, self.D, self.device = num_classes, int(dim), torch.device(device)
        self.sizes = sizes
        self.stride = max(sizes)
        self.data = torch.zeros(self.K, self.stride, self.D, device=self.device)
        self.state = torch.zeros(self.K, 2, dtype=torch.int32, device=self.device)     # rows, head
        self.caps = torch.tensor(sizes, dtype=torch.int32, device=self.device)
        self._ptrs = [0] * self.K                                                       # the reference's queue_ptr
        self.pending = []       # device logs [K][2] = (new keys, rows after) of one-pass calls not yet folded into _ptrs
        self.calls = 0

    @property
    def ptrs(self):
        """the reference's queue_ptr per class; one-pass calls log their enqueue counts on the device and the
        pointers are brought up to date here, when someone asks (one read-back for all pending calls)"""
        self._fold_pending()
        return self._ptrs

    def _fold_pending(self):
        if self.pending:
            logs = torch.stack(self.pending).cpu().tolist()
            self.pending = []
            for log in logs:
                for c in range(self.K):
                    self.note_enqueued(c, int(log[c][1]), int(log[c][0]))

    def log_call(self, keys_log):
        """remember one one-pass call's device log; the list is bounded (one small read-back every PENDING_MAX calls
        when nobody asks for the pointers in between, instead of an ever-growing list of device tensors)"""
        self.pending.append(keys_log)
        if len(self.pending) >= PENDING_MAX:
            self._fold_pending()

    # ---- reference-style views -------------------------------------------------------------
    def host_state(self):
        st = self.state.cpu()
        return st[:, 0].tolist(), st[:, 1].tolist()

    def rows(self, c):
        """logical contents of class c, oldest first (what memobank[c][0] holds in the reference)"""
        rows, head = self.host_state()
        idx = (head[c] + torch.arange(rows[c], device=self.device)) % self.sizes[c]
        return self.data[c].index_select(0, idx)

    @classmethod
    def from_lists(cls, memobank, queue_prtlis, queue_size, dim, device):
        bank = cls(len(memobank), list(queue_size), dim, device)
        for c, q in enumerate(memobank):
            t = q[0]
            if t.shape[0]:
                bank.push(c, t)
            bank._ptrs[c] = int(queue_prtlis[c][0]) if queue_prtlis is not None else bank._ptrs[c]
        return bank

    def to_lists(self, memobank, queue_prtlis):
        for c in range(self.K):
            memobank[c][0] = self.rows(c)
            if queue_prtlis is not None:
                queue_prtlis[c][0] = self.ptrs[c]

    # ---- dequeue_and_enqueue (loss_helper.py:19-36) -------------------------------------------
    def push(self, c, keys):
        lib = _lib.load()
        _ = self.ptrs                        # fold pending one-pass logs in first: pointer updates are ordered
        keys = _f32(keys, self.device)
        m = int(keys.shape[0])
        _lib.check("cmlpl_memobank_push", lib.cmlpl_memobank_push(
            _p(keys) if m else None, m, self.D, _p(self.data[c]), _p(self.state[c]), self.sizes[c],
            _stream(self.device)))
        rows, _ = self.host_state()
        self.note_enqueued(c, rows[c], m)
        return m

    def note_enqueued(self, c, rows_after, m):
        """pointer rule of the reference: `size` once the bank is full, (ptr + m) % size before that"""
        self._ptrs[c] = self.sizes[c] if rows_after >= self.sizes[c] else (self._ptrs[c] + m) % self.sizes[c]


def dequeue_and_enqueue(keys, bank: MemoryBank, c: int):
    """loss_helper.py:19-36 for class c of a device-resident bank; returns the batch size like the reference"""
    return bank.push(c, keys)


class _Scaled(torch.autograd.Function):
    """loss whose gradient with respect to `x` was produced by the same kernels as the value"""

    @staticmethod
    def forward(ctx, x, loss, grad):
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None


def unsupervised_loss(predict, target, percent, pred_teacher):
    """compute_unsupervised_loss (loss_helper.py:242-261).  `target` (int64, 255 = ignore) is edited in place like
    the reference does; returns the loss (differentiable with respect to `predict`)."""
    lib = _lib.load()
    dev = predict.device
    if dev.type != "cuda":
        raise RuntimeError("cmlpl_amd.memobank needs CUDA (ROCm) tensors: there is no CPU fallback")
    B, K = predict.shape
    if target.dtype != torch.int64 or not target.is_contiguous():
        raise TypeError("target must be a contiguous int64 tensor (it is modified in place)")
    p = _f32(predict, dev)
    t = _f32(pred_teacher, dev)
    loss = torch.empty(1, device=dev)
    grad = torch.empty(B, K, device=dev)
    need = lib.cmlpl_unsup_workspace_bytes(B)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check("cmlpl_unsup_loss", lib.cmlpl_unsup_loss(_p(p), _p(target), _p(t), B, K, float(percent), _p(loss),
                                                        _p(grad), _p(ws), need, _stream(dev)))
    return _Scaled.apply(predict, loss[0], grad)


def contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, bank: MemoryBank, rep_teacher,
                         momentum_prototype=None, i_iter=0, draws=None, as_tensors=None):
    """compute_contra_memobank_loss (loss_helper.py:39-219) on a device-resident MemoryBank, in one pass: three
    kernel launches (cmlpl_memobank_loss) and NO host read-back -- which classes are valid, how many keys were
    enqueued and what each loop position means is worked out on the device from the counts.

    Returns (new_keys, loss) or (prototype, new_keys, loss) like the reference.  `new_keys` is a device int32
    tensor [K] on this fast path (the reference's list of ints would need the read-back); with `as_tensors=False`
    -- or when `draws` are injected, which is the parity-test path -- it is the reference's list.
    `draws`, if given, is (anchor_idx, neg_idx): dicts loop position -> int64 index tensors replacing the two
    torch.randint calls (:164, :179); otherwise the indices are drawn in-kernel (Philox,
    keyed by a fresh 64-bit draw from torch's CPU generator per call: see _draw_key)."""
    lib = _lib.load()
    dev = rep.device
    if dev.type != "cuda":
        raise RuntimeError("cmlpl_amd.memobank needs CUDA (ROCm) tensors: there is no CPU fallback")
    if as_tensors is None:
        as_tensors = draws is None
    st = _stream(dev)
    N, D = rep.shape
    K = label_l.shape[1]
    Nl = label_l.shape[0]
    assert bank.K == K and bank.D == D
    Q, NN = NUM_QUERIES, NUM_NEGATIVES
    keep = [_f32(t, dev) for t in (rep, rep_teacher, prob_l, prob_u, label_l, label_u)]
    keep += [_f32(low_mask, dev).reshape(-1), _f32(high_mask, dev).reshape(-1)]
    # (two allocations instead of nine: a call is three launches of ~34 us together, and every torch.empty is ~3 us of host)
    def carve(buf, *shapes):
        out, o = [], 0
        for shp in shapes:
            n = 1
            for d in shp:
                n *= d
            n4 = (n + 3) & ~3                                  # 16-byte aligned pieces
            out.append(buf[o:o + n].view(*shp))
            o += n4
        return out
    isz = lambda *shapes: sum(((torch.Size(sh).numel() + 3) & ~3) for sh in shapes)
    ish = ((K, 3, N), (K, 3), (K, 2), (K, Q))
    fsh = ((K, D), (K, Q), (K, Q, D), (N, D), (1,))
    lists, counts, keys_log, arow = carve(torch.empty(isz(*ish), dtype=torch.int32, device=dev), *ish)
    proto, lossq, ganchor, drep, total = carve(torch.empty(isz(*fsh), dtype=torch.float32, device=dev), *fsh)
    c = _lib.MemobankCall()
    (c.d_rep, c.d_rep_teacher, c.d_prob_l, c.d_prob_u, c.d_label_l, c.d_label_u, c.d_low_mask,
     c.d_high_mask) = [t.data_ptr() for t in keep]
    c.N, c.n_labeled, c.K, c.D, c.queries, c.negatives = N, Nl, K, D, Q, NN
    c.d_bank, c.d_state, c.d_capacity, c.capacity_stride = (bank.data.data_ptr(), bank.state.data_ptr(),
                                                            bank.caps.data_ptr(), bank.stride)
    if draws is not None:      # dense [K][..] tensors indexed by loop position
        a_all = torch.zeros(K, Q, dtype=torch.int64, device=dev)
        n_all = torch.zeros(K, Q * NN, dtype=torch.int64, device=dev)
        for i, t in draws[0].items():
            a_all[i].copy_(t.to(device=dev, dtype=torch.int64))
        for i, t in draws[1].items():
            n_all[i].copy_(t.to(device=dev, dtype=torch.int64))
        keep += [a_all, n_all]
        c.d_anchor_draw, c.d_neg_draw = a_all.data_ptr(), n_all.data_ptr()
    bank.calls += 1
    c.seed, c.call = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, _draw_key()
    prototype = None
    if momentum_prototype is not None:                    # :194-203
        mp = _f32(momentum_prototype, dev).reshape(K, Q, D)
        on = (mp != 0).any().to(torch.int32).reshape(1)    # "not (momentum_prototype == 0).all()", kept on device
        if i_iter == 0 and bool(on.item()):
            raise ZeroDivisionError("division by zero")   # ema = min(1 - 1 / i_iter, 0.999), as the reference raises
        prototype = torch.zeros(K, Q, 1, D, device=dev)
        keep += [mp, on]
        c.d_momentum, c.d_momentum_on, c.d_prototype = mp.data_ptr(), on.data_ptr(), prototype.data_ptr()
        c.ema = min(1 - 1 / i_iter, 0.999) if i_iter else 0.0
    c.temperature = TEMP
    (c.d_lists, c.d_counts, c.d_proto, c.d_keys_log, c.d_lossq, c.d_ganchor, c.d_arow, c.d_drep,
     c.d_total) = [t.data_ptr() for t in (lists, counts, proto, keys_log, lossq, ganchor, arow, drep, total)]
    _lib.check("cmlpl_memobank_loss", lib.cmlpl_memobank_loss(C.byref(c), st))
    bank.log_call(keys_log)                               # pointer bookkeeping is resolved when someone asks
    loss = _Scaled.apply(rep, total[0], drep)
    if draws is not None:       # parity path: injected indices must have been in range (the kernel clamps them)
        cnt, (rows, _) = counts.cpu(), bank.host_state()
        valid = [k for k in range(K) if int(cnt[k, 0]) > 0]
        if len(valid) > 1:
            for i, vc in enumerate(valid):
                if int(cnt[i, 1]) == 0 or rows[vc] == 0:
                    continue
                a_i, n_i = draws[0][i], draws[1][i]
                if int(a_i.max()) >= int(cnt[i, 1]) or int(n_i.max()) >= rows[vc] or int(a_i.min()) < 0 or int(n_i.min()) < 0:
                    raise IndexError("injected draw out of range")
    new_keys = keys_log[:, 0] if as_tensors else keys_log[:, 0].tolist()
    del keep
    return (new_keys, loss) if momentum_prototype is None else (prototype, new_keys, loss)
