#!/usr/bin/env python3
"""Host cost of one training step, eager (cmlpl_train_step: one ctypes call, ten launches) against replayed from the
captured hipGraph (cmlpl_step_graph_launch: one launch call), next to the device time of the step -- at the headline
batch and at the per-rank shard sizes of BASELINE configs[2] / configs[4] on 8 GPUs, where the device time is shortest.
   python scripts/graph_overhead.py            (one GPU; prints a table)
host us/step   = wall time to ENQUEUE a step (the stream kept busy, no synchronisation inside the loop)
device us/step = wall time of N steps including the final synchronisation, divided by N
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cmlpl_amd import HyperParams, NetShape, TrainEngine  # noqa: E402

DEV = "cuda:0"
CASES = [("B2 128+128 (headline)", (103, 11, 11, 103, 9), 128, 128),
         ("B2  64+64  (configs[2] per rank at 8 GPUs)", (103, 11, 11, 103, 9), 64, 64),
         ("B5   8+64  (configs[4] per rank at 8 GPUs)", (48, 15, 15, 48, 20), 8, 64)]
N = 400


def data(shape, nl, nu):
    C, H, W, bands, K = shape
    g = torch.Generator().manual_seed(1)
    return [t.to(DEV) for t in (torch.randn(nl, C, H, W, generator=g), torch.randn(nl, bands, generator=g),
                                torch.randint(0, K, (nl,), generator=g), torch.randn(nu, C, H, W, generator=g),
                                torch.randn(nu, bands, generator=g))]


def measure(run, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        run(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def main():
    print(f"{'case':46s} {'mode':7s} {'host us/step':>13s} {'device us/step':>15s}")
    for name, shape, bt, btu in CASES:
        XP, X, Y, XPu, Xu = data(shape, 8 * bt, 8 * btu)
        eng = TrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=DEV, seed=1088, hist_rows=16)
        eng.init_params_default(1088)
        lab = torch.arange(8 * bt, device=DEV)
        unl = torch.arange(8 * btu, device=DEV)

        def eager(i):
            o = (i % 8)
            eng.step(XP, X, Y, XPu, Xu, 1, i, lab_idx=lab[o * bt:(o + 1) * bt], unl_idx=unl[o * btu:(o + 1) * btu])
        for i in range(20):
            eager(i)
        h, d = measure(eager, N)
        print(f"{name:46s} {'eager':7s} {h:13.1f} {d:15.1f}")
        graph = eng.capture(XP, X, Y, XPu, Xu, lab, unl, bt, btu, capacity=N + 32)
        graph.program([(1, i, (i % 8) * bt, (i % 8) * btu) for i in range(20)])
        for i in range(20):
            graph.launch()
        graph.program([(1, i, (i % 8) * bt, (i % 8) * btu) for i in range(N)])
        h, d = measure(lambda i: graph.launch(), N)
        print(f"{name:46s} {'graph':7s} {h:13.1f} {d:15.1f}")
        graph.close()


if __name__ == "__main__":
    main()
