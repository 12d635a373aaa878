"""train.py's epoch / step loop as TWO real processes (gloo, CPU) around the stand-in engine: the sharding
decisions are taken on the global batch, so both ranks run the same steps -- including the short last batch of an
epoch (44 samples, batch 8+8: five full steps and one of 4+4 -> 2+2 per rank), which round 1's loop turned into
a mis-sized shard -- and loss_hist (train.py:136,274-278) holds every step's row."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_train_loop_child.py")


def _single_process(num_unlabel):
    sys.path.insert(0, ROOT)
    import torch
    import train
    from tests.cpu_dist_engine import CpuLoopEngine

    from cmlpl_amd.distributed import SingleComm as One       # world size 1: every collective is a copy

    args = train.build_parser().parse_args([
        "--synthetic", "B2", "--num_unlabel", str(num_unlabel), "--labeled_batch_size", "8",
        "--unlabeled_batch_size", "8", "--num_epochs", "2", "--print_per_batches", "2", "--no_eval", "--dropout", "0"])
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK") if k in os.environ}
    try:
        return train.main(args, make_engine=lambda shape, bt, btu, hp, ppb: CpuLoopEngine(shape, bt, btu, hp, One(), ppb),
                          device=torch.device("cpu"))
    finally:
        os.environ.update(env)


def test_two_rank_train_loop_equals_one_rank(tmp_path):
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(2, [sys.executable, CHILD, str(tmp_path / "h"), "44"], timeout=900)
    assert rc == 0, out
    h0, h1 = np.load(tmp_path / "h_rank0.npy"), np.load(tmp_path / "h_rank1.npy")
    assert h0.shape == (12, 5) and np.array_equal(h0, h1)                 # 2 epochs x 6 batches; ranks agree
    assert np.all(h0[:, 1] > 0)                                          # every step ran (total_loss row filled)
    lines = [ln for ln in out.splitlines() if ln.startswith("Epoch")]
    assert len(lines) == 6                                               # 3 prints per epoch (print_per_batches 2)
    # the printed numbers are window means of loss_hist, as in train.py:281-289
    first = lines[0]
    assert first.startswith("Epoch 1/2:  2/6 ")
    want = 'loss_contrast= %.2f total_loss = %.4f cls_loss = %.4f con_loss = %.4f acc = %.2f' % (
        h0[0:2, 0].mean(), h0[0:2, 1].mean(), h0[0:2, 2].mean(), h0[0:2, 3].mean(), h0[0:2, 4].mean() * 100)
    assert want in first, (want, first)
    # same loop in one process on the same global batches
    ref = _single_process(44)
    assert ref.shape == h0.shape
    assert np.allclose(h0, ref, rtol=2e-4, atol=1e-5), np.abs(h0 - ref).max()


def test_odd_short_batch_is_cut_to_equal_shards(tmp_path):
    """45 samples: the last batch has 5+5 rows -> 2+2 per rank, one row of each dropped; both ranks still take
    the same decisions and finish."""
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(2, [sys.executable, CHILD, str(tmp_path / "h"), "45"], timeout=900)
    assert rc == 0, out
    h0, h1 = np.load(tmp_path / "h_rank0.npy"), np.load(tmp_path / "h_rank1.npy")
    assert h0.shape == (12, 5) and np.array_equal(h0, h1) and np.all(h0[:, 1] > 0)
