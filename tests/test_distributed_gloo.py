"""CPU, world_size 2, gloo: the real multi-process wiring of the sharded step -- drive_step +
TorchDistComm from cmlpl_amd.distributed around the CPU stand-in engine -- must reproduce the
single-process oracle on the same GLOBAL batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import cmlpl_oracle as O
    from cmlpl_amd.distributed import TorchDistComm, drive_step      # host logic only: imports without a GPU
    from tests.cpu_dist_engine import CpuDistEngine
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape = O.NetShape(103, 11, 11, 103, 9)
    hp = O.HyperParams()
    bt, btu = 16, 24
    p0, p1 = O.closed_form_params(shape, 41), O.closed_form_params(shape, 42)
    eng = CpuDistEngine(shape, bt // world, btu // world, hp, world, rank, p0, p1)
    comm = TorchDistComm()
    ref = O.StepState.create(shape, p0, p1, bt, hp) if rank == 0 else None
    bl, bul = bt // world, btu // world
    ls, us = slice(rank * bl, (rank + 1) * bl), slice(rank * bul, (rank + 1) * bul)
    res = []
    for s in range(3):
        b = O.synthetic_batch(shape, bt, btu, 700 + s, separable=1.0)
        nz = b["noise"]
        noise = [nz[0][ls], nz[1][ls], nz[2][ls], nz[3][ls], nz[4][us], nz[5][us], nz[6][us], nz[7][us]]
        dm = [torch.cat([m[ls], m[bt:][us]]) for m in b["dropmask"]]
        drive_step(eng, comm, b["XPl"][ls], b["Xl"][ls], b["Y"][ls], b["XPu"][us], b["Xu"][us], 1, s, noise, dm)
        sc = eng.scalars.clone()
        comm.all_reduce(sc)
        if rank == 0:
            r = O.train_step(ref, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"], 1, s, hp)
            want = r["hist"] + [float(r["total_w"]), float(r["cls_w"]), float(r["con_w"]), float(r["ctr_w"])]
            res.append((sc[:9].tolist(), want))
    if rank == 0:
        perr = max(float((eng.params[n][k] - ref.params[n][k]).abs().max()) for n in range(2) for k in O.LIVE_KEYS)
        berr = max(float((eng.bank_feats[i] - ref.bank_feats[i]).abs().max()) for i in range(2))
        out.put((res, perr, berr, eng.ptr, ref.ptr))
    # all replicas hold identical parameters
    flat = torch.cat([eng.params[n][k].reshape(-1) for n in range(2) for k in O.LIVE_KEYS])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(o, other[0]) for o in other)
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process_oracle():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    import queue, time
    t0, got = time.time(), None
    while got is None:
        try:
            got = out.get(timeout=2)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > 600:
                for p in procs:
                    p.terminate()
                pytest.fail("a rank died or timed out: " + str([p.exitcode for p in procs]))
    res, perr, berr, ptr, ptr_ref = got
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for got, want in res:
        assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (got, want)
    assert perr < 3e-5 and berr < 1e-5, (perr, berr)
    assert ptr == ptr_ref
