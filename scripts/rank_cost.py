"""Per-rank device cost of the data-parallel step at world size W, measured on ONE GPU: rank 0 of W is emulated
(local 128+128 rows, global W x that, bank 10 x bt_global rows); the exchanges are faked by replicating the rank's
own buffers, so collective time is NOT included -- this isolates how the per-rank kernels grow with W.
    python scripts/rank_cost.py 8                      weak scaling: B2, 128 + 128 rows per rank
    python scripts/rank_cost.py 8 B3 64 64             strong scaling: BASELINE configs[2] (global 512 + 512 over 8 ranks)
    python scripts/rank_cost.py 8 B5 8 64              strong scaling: BASELINE configs[4] (global 64 + 512 over 8 ranks)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import WORKLOADS, synth
from cmlpl_amd import HyperParams, NetShape, _lib
from cmlpl_amd.distributed import DistTrainEngine

DEV = torch.device("cuda:0")


class FakeComm:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank


def step(e, b, i):
    W = e.comm.world
    e.stage_spectral(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, i)
    for stage in e.STAGES:
        if stage != "spectral":
            getattr(e, "stage_" + stage)()
        for spec in e.exchange_after(stage):
            kind = spec[0]
            if kind == "all_gather":
                spec[1].view(W, -1).copy_(spec[2].reshape(1, -1).expand(W, -1))
            elif kind == "reduce_scatter":
                spec[1].copy_(spec[2].chunk(W, dim=0)[0])


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    wl = sys.argv[2] if len(sys.argv) > 2 else "B2"
    bt = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    btu = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    shape = WORKLOADS[wl]
    e = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=DEV, seed=1088, comm=FakeComm(W, 0))
    e.init_params_default(1088)
    b = synth(shape, bt, btu, 1, DEV)
    lib = _lib.load()
    for i in range(10):
        step(e, b, i)
    torch.cuda.synchronize()
    nk = len(_lib.KERNEL_NAMES)
    ms, cnt = (C.c_double * nk)(), (C.c_int64 * nk)()
    _lib.check("cmlpl_timing_begin", lib.cmlpl_timing_begin(0xFFFFFFFF, 40 * 30))
    for i in range(30):
        step(e, b, 10 + i)
    torch.cuda.synchronize()
    _lib.check("cmlpl_timing_end", lib.cmlpl_timing_end(ms, cnt))
    tot = sum(ms[i] for i in range(nk))
    print(f"W={W} {wl} {bt}+{btu} rows per rank ({2 * (bt + btu)} sample-net workgroups): Q={e.Q} bank rows, per-rank kernel "
          f"time {tot / 30 * 1e3:.1f} us/step (hipEvent pairs, no collectives)")
    for i, nm in enumerate(_lib.KERNEL_NAMES):
        if cnt[i] and ms[i] / 30 * 1e3 >= 3.0:
            print(f"  {nm:12s} {ms[i] / cnt[i] * 1e3:9.1f} us/launch x{cnt[i] // 30}")


if __name__ == "__main__":
    main()
