"""Measurements for the SURVEY.md 8(f) rows built beyond the training step (one MI355X, synthetic data):
  N1 whole-image inference (eval forward, batch 512)         -> patches/s
  N3 on-device patch extraction (mirror pad + w x w gather)  -> GB/s written (HBM-bound gather)
  N4 tools.models.ContrastiveLoss (NT-Xent) forward+backward -> us / call
  N2 loss_helper: memory-bank InfoNCE and entropy-filtered CE -> us / call
    python scripts/bench_next_rows.py > gpurun_out/next_rows.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

DEV = torch.device("cuda:0")


def timeit(fn, warm=5, reps=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def n1():
    from tools.models import BaseNet2
    net = BaseNet2(num_features=103, dropout=0.8, num_classes=9, in_channels=103, window=11).to(DEV).eval()
    x = torch.randn(512, 103, 11, 11, device=DEV)
    y = torch.randn(512, 103, device=DEV)
    with torch.no_grad():
        dt = timeit(lambda: net(x, y))
    print(f"N1 whole-image inference, B2 shape, batch 512: {dt * 1e3:.3f} ms/batch = {512 / dt / 1e3:.0f} k patches/s "
          f"(PaviaU's 207,400 pixels: {207400 / (512 / dt):.2f} s)")


def n3():
    from cmlpl_amd.patches import extract_patches
    H, W, C, w = 610, 340, 103, 11            # PaviaU cube
    cube = torch.randn(H, W, C, device=DEV)
    idx = torch.randint(0, H * W, (8192,), device=DEV)
    out = torch.empty(8192, C, w, w, device=DEV)
    dt = timeit(lambda: extract_patches(cube, idx, w, out=out))
    gb = out.numel() * 4 / 1e9
    print(f"N3 patch extraction, 8192 patches of {w}x{w}x{C} from a {H}x{W}x{C} cube: {dt * 1e6:.1f} us = "
          f"{gb / dt:.0f} GB/s written, random order (reads + writes = 2x that much memory traffic)")
    ids = torch.sort(idx).values              # the same patches in raster order (what whole-image inference asks for)
    dt = timeit(lambda: extract_patches(cube, ids, w, out=out))
    print(f"N3 patch extraction, the same 8192 patches in raster order: {dt * 1e6:.1f} us = {gb / dt:.0f} GB/s written "
          f"(overlapping windows re-read from each XCD's L2)")


def n4():
    from tools.models import ContrastiveLoss
    for B, D in ((128, 1024), (256, 1024), (512, 1024)):
        ei = torch.randn(B, D, device=DEV, requires_grad=True)
        ej = torch.randn(B, D, device=DEV, requires_grad=True)
        crit = ContrastiveLoss(B, device="cuda", temperature=0.5)

        def step():
            ei.grad = ej.grad = None
            crit(ei, ej).backward()
        dt = min(timeit(step, warm=30, reps=300) for _ in range(5))     # (host-bound call: short windows catch the box's CPU hiccups)
        print(f"N4 NT-Xent forward+backward, B={B}, D={D}: {dt * 1e6:.1f} us/call wall")


def n2():
    import loss_helper as LH
    from cmlpl_amd.memobank import MemoryBank
    rng = np.random.Generator(np.random.PCG64(5))
    Nl = Nu = 128
    N, D, K, cap = Nl + Nu, 256, 9, 30000
    onehot = torch.eye(K)
    yl, yu = torch.from_numpy(rng.integers(0, K, Nl)), torch.from_numpy(rng.integers(0, K, Nu))
    args = dict(label_l=onehot[yl].to(DEV), label_u=onehot[yu].to(DEV),
                prob_l=torch.softmax(torch.randn(Nl, K) + 2 * onehot[yl], 1).to(DEV),
                prob_u=torch.softmax(torch.randn(Nu, K) + onehot[yu], 1).to(DEV),
                low_mask=(torch.rand(N, 1) < 0.7).float().to(DEV), high_mask=(torch.rand(N, 1) < 0.6).float().to(DEV))
    bank = MemoryBank(K, cap, D, DEV)
    for c in range(K):
        bank.push(c, torch.randn(20000, D, device=DEV))
    rep = torch.randn(N, D, device=DEV, requires_grad=True)
    rep_t = torch.randn(N, D, device=DEV)

    def step():
        rep.grad = None
        _, loss = LH.compute_contra_memobank_loss(rep, args["label_l"], args["label_u"], args["prob_l"], args["prob_u"],
                                                  args["low_mask"], args["high_mask"], bank, None, None, rep_t)
        loss.backward()
    # wall time of a call (Python + autograd around three launches): best of five batches of 100 calls.  Per-kernel
    # device times are NOT printed here (a constant in a program's output goes stale): scripts/kstats_next_rows.sh
    # takes them from rocprofv3 for the same calls.
    dt = min(timeit(step, warm=10, reps=100) for _ in range(5))
    gathered = K * 256 * 51 * D * 4 / 1e9          # every key row crosses HBM once (the backward re-uses the registers)
    print(f"N2 compute_contra_memobank_loss fwd+bwd, N={N}, D={D}, K={K}, banks of 20000+ rows (cap {cap}): "
          f"{dt * 1e6:.0f} us/call wall, no host read-back; {gathered:.2f} GB of key gathers per call")
    B = 4096
    pred = torch.randn(B, K, device=DEV, requires_grad=True)
    teach = torch.randn(B, K, device=DEV) * 3
    tgt0 = torch.randint(0, K, (B,), device=DEV)

    def ustep():
        pred.grad = None
        LH.compute_unsupervised_loss(pred, tgt0.clone(), 80.0, teach).backward()
    dt = min(timeit(ustep, warm=10, reps=100) for _ in range(3))
    print(f"N2 compute_unsupervised_loss fwd+bwd, B={B}, K={K}: {dt * 1e6:.0f} us/call wall")


if __name__ == "__main__":
    want = sys.argv[1:]                       # e.g. `bench_next_rows.py n2 n3`; default: all
    for f in (n1, n3, n4, n2):
        if not want or f.__name__ in want:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()          # a section's big buffers must not shape the next one's allocations
            f()
