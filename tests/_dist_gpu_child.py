"""Child of tests/test_gpu_multiprocess.py: one rank of a REAL multi-process data-parallel job on ONE GPU (every rank
uses cuda:0; the collectives run on gloo because RCCL refuses two ranks on one device).  The sharded step --
DistTrainEngine + TorchDistComm, the shipped classes -- runs W ranks on shards of a global batch; rank 0 also runs
the plain TrainEngine on the whole batch and compares."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cmlpl_amd import HyperParams, NetShape, TrainEngine  # noqa: E402
from cmlpl_amd.distributed import DistTrainEngine  # noqa: E402
from oracle import cmlpl_oracle as O  # noqa: E402  (input generators only)

rank, W = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
shape = O.NetShape(103, 11, 11, 103, 9)
bt, btu, steps = (int(os.environ.get(k, v)) for k, v in (("CMLPL_TEST_BT", 32), ("CMLPL_TEST_BTU", 64), ("CMLPL_TEST_STEPS", 3)))
hp = HyperParams()
p0, p1 = O.closed_form_params(shape, 51), O.closed_form_params(shape, 52)
if os.environ.get("CMLPL_TEST_GRAPH") == "1":
    # DistStepGraph.launch() itself -- seven graph launches with the REAL collectives between them, two of them asynchronous
    # (what `train.py --graph` runs on several GPUs) -- against the eager drive_step on the same process group: bit for bit
    bl, bul = bt // W, btu // W
    g = torch.Generator().manual_seed(9)
    NL, NU = 4 * bt, 4 * btu
    d = lambda t: t.to(dev).contiguous()
    XP, X, Y = d(torch.randn(NL, 103, 11, 11, generator=g)), d(torch.randn(NL, 103, generator=g)), d(torch.randint(0, 9, (NL,), generator=g))
    XPu, Xu = d(torch.randn(NU, 103, 11, 11, generator=g)), d(torch.randn(NU, 103, generator=g))
    lp, up = d(torch.randperm(NL, generator=g)), d(torch.randperm(NU, generator=g))
    sched = [(0, 16), (0, 17), (0, 18), (1, 0), (1, 1)]          # crosses the smoothing gate and an epoch boundary
    offs = [((k % 4) * bt + rank * bl, (k % 4) * btu + rank * bul) for k in range(len(sched))]
    engs = []
    for _ in range(2):
        e = DistTrainEngine(NetShape(103, 11, 11, 103, 9), bl, bul, hp, device=dev, seed=5, hist_rows=8)
        e.load_state_dict(0, p0); e.load_state_dict(1, p1)
        engs.append(e)
    ea, eb = engs
    def eager(e, k):
        lo, uo = offs[k]
        e.step(XP, X, Y, XPu, Xu, sched[k][0], sched[k][1], lab_idx=lp[lo:lo + bl], unl_idx=up[uo:uo + bul])
    for k in range(len(sched)):
        eager(ea, k)
    eager(eb, 0)
    gr = eb.capture(XP, X, Y, XPu, Xu, lp, up, bl, bul, capacity=8)
    gr.program([(sched[k][0], sched[k][1], offs[k][0], offs[k][1]) for k in range(1, len(sched))])
    for k in range(1, len(sched)):
        gr.launch()
    torch.cuda.synchronize()
    for name in ("params", "m", "v", "grads", "bank_feats", "bank_probs", "scalar_hist"):
        x, y = getattr(ea, name), getattr(eb, name)
        assert torch.equal(x, y), f"rank {rank}: {name} differs, max |d| = {(x - y).abs().max().item():.3e}"
    assert ea.ptr == eb.ptr and ea.adam_t == eb.adam_t and ea.step_count == eb.step_count
    assert torch.isfinite(eb.scalar_hist[:len(sched)]).all()
    gr.close()
    if rank == 0:
        print(f"OK graph world={W} steps={len(sched)}")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
eng = DistTrainEngine(NetShape(103, 11, 11, 103, 9), bt // W, btu // W, hp, device=dev, seed=5)
eng.load_state_dict(0, p0); eng.load_state_dict(1, p1)
ref = None
if rank == 0:
    ref = TrainEngine(NetShape(103, 11, 11, 103, 9), bt, btu, hp, device=dev, seed=5)
    ref.load_state_dict(0, p0); ref.load_state_dict(1, p1)
bl, bul = bt // W, btu // W
ls, us = slice(rank * bl, (rank + 1) * bl), slice(rank * bul, (rank + 1) * bul)
worst = 0.0
for s in range(steps):
    b = O.synthetic_batch(shape, bt, btu, 800 + s, separable=1.0)
    d = lambda t: t.to(dev).contiguous()
    nz = b["noise"]
    noise = [d(nz[0][ls]), d(nz[1][ls]), d(nz[2][ls]), d(nz[3][ls]), d(nz[4][us]), d(nz[5][us]), d(nz[6][us]), d(nz[7][us])]
    dm = torch.stack([torch.cat([m[ls], m[bt:][us]]) for m in b["dropmask"]]).to(dev).contiguous()
    if os.environ.get("CMLPL_TEST_BYIDX") == "1":
        # the shard BY INDEX: every rank holds the whole (shuffled) splits and takes its rows through index lists
        g = torch.Generator().manual_seed(40 + s)
        pl, pu = torch.randperm(bt, generator=g), torch.randperm(btu, generator=g)
        inv_l, inv_u = torch.argsort(pl), torch.argsort(pu)            # row r of the batch sits at inv[r] of the split
        eng.step(d(b["XPl"][pl]), d(b["Xl"][pl]), d(b["Y"][pl]), d(b["XPu"][pu]), d(b["Xu"][pu]), 1, s, noise=noise,
                 dropmask=dm, lab_idx=inv_l[ls].to(dev).contiguous(), unl_idx=inv_u[us].to(dev).contiguous())
    else:
        eng.step(d(b["XPl"][ls]), d(b["Xl"][ls]), d(b["Y"][ls]), d(b["XPu"][us]), d(b["Xu"][us]), 1, s, noise=noise, dropmask=dm)
    got = eng.read_scalars()                       # all-reduced over the ranks
    if rank == 0:
        ref.step(d(b["XPl"]), d(b["Xl"]), d(b["Y"]), d(b["XPu"]), d(b["Xu"]), 1, s, noise=[d(t) for t in nz],
                 dropmask=torch.stack(b["dropmask"]).to(dev).contiguous())
        want = ref.read_scalars()
        for k in ("ctr_s", "total_s", "cls_s", "con_s", "acc", "total_w", "cls_w", "con_w"):
            assert abs(got[k] - want[k]) <= 1e-5 * abs(want[k]) + 1e-6, (s, k, got[k], want[k])
        assert [got[k] for k in ("n_mask_w", "n_mask_s", "n_pos", "n_neg")] == [want[k] for k in ("n_mask_w", "n_mask_s", "n_pos", "n_neg")]
        live = eng.live
        for net in range(2):
            g, r = eng.grads[net], ref.grads[net, :live]
            err = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-9)
            worst = max(worst, err)
            assert err < 2e-4, (s, net, err)
        assert eng.ptr == ref.ptr
        for i in range(2):
            assert float((eng.bank_feats[i] - ref.bank_feats[i]).abs().max()) < 1e-5
            assert float((eng.bank_probs[i] - ref.bank_probs[i]).abs().max()) < 1e-5
    # replicas hold identical parameters after this step's update
    psum = eng.params.double().sum().reshape(1).cpu()
    lst = [torch.zeros_like(psum) for _ in range(W)]
    dist.all_gather(lst, psum)
    assert all(float(x) == float(lst[0]) for x in lst), lst
    # every step is compared from EQUAL states (tests/test_gpu_distributed.py explains why): all ranks continue from
    # the single-process engine's parameters, Adam moments and banks
    for name in ("params", "m", "v"):
        t = getattr(eng, name)
        if rank == 0:
            t.copy_(getattr(ref, name))
        dist.broadcast(t, 0)
    for i in range(2):
        for bank in (eng.bank_feats, eng.bank_probs):
            if rank == 0:
                bank[i].copy_((ref.bank_feats if bank is eng.bank_feats else ref.bank_probs)[i])
            dist.broadcast(bank[i], 0)
    eng._packed_dirty = True
if rank == 0:
    print(f"OK world={W} steps={steps} worst_grad_rel_err={worst:.2e}")
dist.barrier()
dist.destroy_process_group()
