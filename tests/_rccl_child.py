"""Child of tests/test_gpu_rccl_comm.py: one rank (RCCL at world size 1: this pool has one GPU per box) runs the sharded
step with its four REAL collectives twice -- through torch.distributed (TorchDistComm) and straight on librccl (RcclComm)
-- and both must leave bit-identical state; also a direct check of the three collectives against their definitions."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from cmlpl_amd import HyperParams, NetShape  # noqa: E402
from cmlpl_amd.distributed import DistTrainEngine, NoOpComm, TorchDistComm  # noqa: E402
from cmlpl_amd.rccl_comm import RcclComm  # noqa: E402

rc = RcclComm(dev)
# the collectives by themselves (world 1: gather = copy, reduce-scatter = copy, all-reduce = identity), sync and async
a = torch.randn(1000, device=dev); o = torch.zeros(1000, device=dev)
rc.all_gather(o, a); assert torch.equal(o, a)
o.zero_(); h = rc.all_gather(o, a, async_op=True); h.wait(); assert torch.equal(o, a)
o.zero_(); h = rc.reduce_scatter(o, a, async_op=True); h.wait(); assert torch.equal(o, a)
b = a.clone(); rc.all_reduce(b); assert torch.equal(b, a)

shape = (103, 11, 11, 103, 9)
bt, btu = 24, 40
g = torch.Generator().manual_seed(3)
d = lambda t: t.to(dev).contiguous()
XP, X, Y = d(torch.randn(bt, 103, 11, 11, generator=g)), d(torch.randn(bt, 103, generator=g)), d(torch.randint(0, 9, (bt,), generator=g))
XPu, Xu = d(torch.randn(btu, 103, 11, 11, generator=g)), d(torch.randn(btu, 103, generator=g))
states = []
# the Python-driven step over torch.distributed | the ONE-call native step (cmlpl_dist_step) over RCCL with the two large
# exchanges forked onto the side stream | the same with all four in order | the Python-driven step over RcclComm | one rank with the collectives aliased away: native | Python-driven
for comm, native, asyn in ((TorchDistComm(), False, None), (rc, True, True), (rc, True, False), (rc, False, None),
                           (NoOpComm(), True, None), (NoOpComm(), False, None)):
    e = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=11, comm=comm, alias_single=False, hist_rows=4)
    e.native_step, e.async_exchanges = native, asyn
    assert (e._native_comm() is not False) == native
    e.init_params_default(5)
    for s in range(4):
        e.step(XP, X, Y, XPu, Xu, 1, s, apply_update=(s != 2))
    torch.cuda.synchronize()
    states.append([t.clone() for t in (e.params, e.m, e.v, e.grads, e.bank_feats, e.bank_probs, e.scalar_hist)])
    assert torch.isfinite(e.scalar_hist).all()
for k in range(1, len(states)):
    for i, (x, y) in enumerate(zip(states[0], states[k])):
        assert torch.equal(x, y), f"state tensor {i} of variant {k} differs from the torch.distributed path: {(x - y).abs().max().item():.3e}"
rc.close()
# what an engine picks by itself on an RCCL process group: the direct communicator after its checked start-up ...
from cmlpl_amd.distributed import pick_comm  # noqa: E402
picked = pick_comm(dev)
assert isinstance(picked, RcclComm), type(picked)
e = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=11, alias_single=False)
assert isinstance(e.comm, RcclComm) and e._native_comm() is not False
picked.close()
os.environ["CMLPL_DIST_COMM"] = "torch"                  # ... unless told otherwise
assert isinstance(pick_comm(dev), TorchDistComm)
print("OK rccl")
dist.destroy_process_group()
