"""Drop-in for the reference's loss_helper.py (the three functions SURVEY.md 8f N2 names), same names, arguments
and return values, on MI355X through cmlpl_amd's C ABI.

`memobank` may be a cmlpl_amd.memobank.MemoryBank (device-resident rings, the fast path) or the reference's own
structure -- a list, per class, of one-element lists holding a [m, D] tensor, with `queue_prtlis` / `queue_size`
beside it; in the latter case the lists are uploaded, updated and written back (cat + [-size:] semantics,
loss_helper.py:19-36)."""
import torch

from cmlpl_amd.memobank import MemoryBank, contra_memobank_loss, unsupervised_loss

__all__ = ["dequeue_and_enqueue", "compute_contra_memobank_loss", "compute_unsupervised_loss", "MemoryBank"]


def dequeue_and_enqueue(keys, queue, queue_ptr, queue_size):
    """loss_helper.py:19-36.  queue: one-element list holding the class's [m, D] tensor; queue_ptr: 1-element tensor."""
    dev = keys.device if keys.is_cuda else torch.device("cuda")
    D = keys.shape[1] if keys.dim() == 2 else queue[0].shape[1]
    bank = MemoryBank.from_lists([queue], [queue_ptr], [queue_size], D, dev)
    n = bank.push(0, keys.detach())
    queue[0] = bank.rows(0)
    queue_ptr[0] = bank.ptrs[0]
    return n


def compute_contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, memobank, queue_prtlis,
                                 queue_size, rep_teacher, momentum_prototype=None, i_iter=0, _draws=None):
    """loss_helper.py:39-219"""
    if isinstance(memobank, MemoryBank):
        return contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, memobank, rep_teacher,
                                    momentum_prototype, i_iter, _draws)
    bank = MemoryBank.from_lists(memobank, queue_prtlis, queue_size, rep.shape[1], rep.device)
    out = contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, bank, rep_teacher,
                               momentum_prototype, i_iter, _draws, as_tensors=False)
    bank.to_lists(memobank, queue_prtlis)
    return out


def compute_unsupervised_loss(predict, target, percent, pred_teacher):
    """loss_helper.py:242-261 (target is modified in place, like the reference)"""
    return unsupervised_loss(predict, target, percent, pred_teacher)
