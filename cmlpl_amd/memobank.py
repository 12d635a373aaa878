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
        self.ptrs = [0] * self.K                                                        # the reference's queue_ptr

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
            bank.ptrs[c] = int(queue_prtlis[c][0]) if queue_prtlis is not None else bank.ptrs[c]
        return bank

    def to_lists(self, memobank, queue_prtlis):
        for c in range(self.K):
            memobank[c][0] = self.rows(c)
            if queue_prtlis is not None:
                queue_prtlis[c][0] = self.ptrs[c]

    # ---- dequeue_and_enqueue (loss_helper.py:19-36) -------------------------------------------
    def push(self, c, keys):
        lib = _lib.load()
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
        self.ptrs[c] = self.sizes[c] if rows_after >= self.sizes[c] else (self.ptrs[c] + m) % self.sizes[c]


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
                         momentum_prototype=None, i_iter=0, draws=None):
    """compute_contra_memobank_loss (loss_helper.py:39-219) on a device-resident MemoryBank.

    Returns (new_keys, loss) or (prototype, new_keys, loss) like the reference.  `draws`, if given, is
    (anchor_idx, neg_idx): dicts loop position -> int64 index tensors replacing the two torch.randint calls
    (:164, :179); otherwise they are drawn with torch.randint on the device."""
    lib = _lib.load()
    dev = rep.device
    if dev.type != "cuda":
        raise RuntimeError("cmlpl_amd.memobank needs CUDA (ROCm) tensors: there is no CPU fallback")
    st = _stream(dev)
    N, D = rep.shape
    K = label_l.shape[1]
    Nl = label_l.shape[0]
    assert bank.K == K and bank.D == D
    repf = _f32(rep, dev)
    rept = _f32(rep_teacher, dev)
    label = torch.cat((_f32(label_l, dev), _f32(label_u, dev)), dim=0).contiguous()
    prob = torch.cat((_f32(prob_l, dev), _f32(prob_u, dev)), dim=0).contiguous()
    lowm = _f32(low_mask, dev).reshape(-1)
    highm = _f32(high_mask, dev).reshape(-1)
    lists = torch.empty(K, 3, N, dtype=torch.int32, device=dev)
    counts = torch.empty(K, 3, dtype=torch.int32, device=dev)
    proto = torch.empty(K, D, device=dev)
    _lib.check("cmlpl_memobank_select", lib.cmlpl_memobank_select(_p(prob), _p(label), _p(lowm), _p(highm), N, Nl, K,
                                                                  _p(lists), _p(counts), st))
    _lib.check("cmlpl_memobank_proto", lib.cmlpl_memobank_proto(_p(rept), N, D, K, _p(lists), _p(counts), _p(proto), st))
    _lib.check("cmlpl_memobank_enqueue", lib.cmlpl_memobank_enqueue(_p(rept), N, D, K, _p(lists), _p(counts),
                                                                    _p(bank.data), _p(bank.state), _p(bank.caps),
                                                                    bank.stride, st))
    cnt = counts.cpu()                                    # the reference's .item() calls (:135-137)
    rows, head = bank.host_state()
    new_keys = [int(cnt[c, 2]) for c in range(K)]
    for c in range(K):
        bank.note_enqueued(c, rows[c], new_keys[c])
    valid = [c for c in range(K) if int(cnt[c, 0]) > 0]
    if len(valid) <= 1:                                   # :139-145
        loss = 0.0 * rep.sum()
        return (new_keys, loss) if momentum_prototype is None else (momentum_prototype, new_keys, loss)

    valid_seg = len(valid)
    prototype = torch.zeros(K, NUM_QUERIES, 1, D, device=dev) if momentum_prototype is not None else None
    drep = torch.zeros(N, D, device=dev)
    lossq = torch.zeros(valid_seg, NUM_QUERIES, device=dev)
    ganchor = torch.empty(NUM_QUERIES, D, device=dev)
    keep = []                                             # tensors the enqueued kernels read
    for i in range(valid_seg):
        vc = valid[i]
        pool_rows = int(cnt[i, 1])                        # position i's anchor pool (the reference's quirk, :158-168)
        if pool_rows == 0 or rows[vc] == 0:
            continue
        if draws is not None:
            a_idx = draws[0][i].to(device=dev, dtype=torch.int64).contiguous()
            n_idx = draws[1][i].to(device=dev, dtype=torch.int64).contiguous()
            if int(a_idx.max()) >= pool_rows or int(n_idx.max()) >= rows[vc] or int(a_idx.min()) < 0 or int(n_idx.min()) < 0:
                raise IndexError("injected draw out of range")
        else:
            a_idx = torch.randint(pool_rows, (NUM_QUERIES,), device=dev)
            n_idx = torch.randint(rows[vc], (NUM_QUERIES * NUM_NEGATIVES,), device=dev)
        pos = proto[i]                                    # [D], shared by all queries
        qstride = 0
        if momentum_prototype is not None:                # :194-203
            mp = momentum_prototype.to(dev)
            posq = pos.view(1, 1, D).repeat(NUM_QUERIES, 1, 1)
            if not bool((mp == 0).all()):
                ema = min(1 - 1 / i_iter, 0.999)
                posq = (1 - ema) * posq + ema * mp[vc]
            prototype[vc] = posq
            pos = posq.reshape(NUM_QUERIES, D).contiguous()
            qstride = D
        keep += [a_idx, n_idx, pos]
        _lib.check("cmlpl_memobank_infonce", lib.cmlpl_memobank_infonce(
            _p(repf), N, D, _p(lists[i, 1]), pool_rows, _p(a_idx), _p(pos), qstride, _p(bank.data[vc]),
            bank.sizes[vc], rows[vc], head[vc], _p(n_idx), NUM_QUERIES, NUM_NEGATIVES, TEMP, 1.0 / valid_seg,
            _p(lossq[i]), _p(ganchor), _p(drep), st))
    total = torch.empty(1, device=dev)
    _lib.check("cmlpl_memobank_sum", lib.cmlpl_memobank_sum(_p(lossq), valid_seg * NUM_QUERIES, _p(total), st))
    loss = _Scaled.apply(rep, total[0], drep)
    del keep
    return (new_keys, loss) if momentum_prototype is None else (prototype, new_keys, loss)
