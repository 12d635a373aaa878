"""GPU: the drop-in driver `train.py` end to end on synthetic PaviaU-shaped patches -- reference command line
(train.py:356-379), printed line (train.py:281-289: means of loss_hist over the last print_per_batches steps),
loss_hist of every step (train.py:136,274-278), whole-image inference after training (train.py:291-306)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"Epoch (\d+)/(\d+):  (\d+)/(\d+) loss_contrast= ([-\d.naninf]+) total_loss = ([-\d.naninf]+) "
                  r"cls_loss = ([-\d.naninf]+) con_loss = ([-\d.naninf]+) acc = ([-\d.naninf]+)")


def test_train_py_synthetic_epoch(tmp_path):
    hist_path = str(tmp_path / "hist.npy")
    # 300 samples, batch 128+128: 3 batches per epoch (128, 128, 44 -- the short last batch runs, as in the reference)
    r = subprocess.run([sys.executable, "train.py", "--synthetic", "B2", "--num_unlabel", "300", "--num_epochs", "2",
                        "--print_per_batches", "2", "--save_loss_hist", hist_path],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    hist = np.load(hist_path)
    assert hist.shape == (6, 5) and np.isfinite(hist).all() and (hist[:, 1] > 0).all()
    lines = [LINE.search(ln) for ln in r.stdout.splitlines() if ln.startswith("Epoch")]
    assert len(lines) == 2 and all(lines)                     # batch 2 of 3 in each epoch
    for m, idx in zip(lines, (1, 4)):                         # index_i of the printing step
        assert (int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))) == (idx // 3 + 1, 2, 2, 3)
        w = hist[idx - 1:idx + 1]
        want = ("%.2f" % w[:, 0].mean(), "%.4f" % w[:, 1].mean(), "%.4f" % w[:, 2].mean(), "%.4f" % w[:, 3].mean(),
                "%.2f" % (w[:, 4].mean() * 100))
        assert tuple(m.group(i) for i in range(5, 10)) == want, (m.group(0), want)
    assert "training: 6 steps" in r.stdout
    assert r.stdout.count("Result:") == 2 and "OA1=" in r.stdout and "AA=" in r.stdout   # both networks evaluated


def test_train_py_graph_mode_logs_the_same_rows(tmp_path):
    """`--graph`: full batches replayed from the captured step, the short last batch of every epoch (another shape) and
    the very first step run eagerly -- loss_hist must equal the eager run's bit for bit (same kernels, same in-kernel
    random streams, the per-step scalars read from the device table instead of launch arguments)."""
    hists = []
    for extra in ([], ["--graph"]):
        path = str(tmp_path / f"hist{len(extra)}.npy")
        r = subprocess.run([sys.executable, "train.py", "--synthetic", "B2", "--num_unlabel", "700", "--num_epochs", "3",
                            "--print_per_batches", "4", "--no_eval", "--save_loss_hist", path] + extra,
                           cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "training: 18 steps" in r.stdout          # 700 samples, 128 + 128: five full batches and one of 60 per epoch
        hists.append(np.load(path))
    assert hists[0].shape == (18, 5) and np.isfinite(hists[0]).all()
    assert np.array_equal(hists[0], hists[1])
