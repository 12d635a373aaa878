"""Do the one-call sharded step's two asynchronous collectives overlap the convolutions?  Wall per step (median of five
windows) of cmlpl_dist_step over RcclComm at world size 1, issued on torch's default (null) stream and on a created
stream, against the aliased step (no collectives)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from cmlpl_amd import NetShape, HyperParams
from cmlpl_amd.distributed import DistTrainEngine, NoOpComm
from cmlpl_amd.rccl_comm import RcclComm
from bench import synth, WORKLOADS
rc = RcclComm(dev)
K = 100
for wl, bt, btu in (("B2", 64, 64), ("B2", 128, 128)):
    shape = WORKLOADS[wl]
    b = synth(shape, bt, btu, 1, dev)
    for name, comm, stream in (("aliased, default stream", NoOpComm(), None), ("RCCL, default stream", rc, None),
                               ("aliased, created stream", NoOpComm(), torch.cuda.Stream(dev)), ("RCCL, created stream", rc, torch.cuda.Stream(dev))):
        eng = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088, comm=comm, alias_single=False)
        eng.init_params_default(1088)
        torch.cuda.synchronize()
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev))
        with ctx:
            for i in range(20):
                eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, i)
            torch.cuda.synchronize()
            ws = []
            for w in range(5):
                t0 = time.perf_counter()
                for i in range(K):
                    eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, 20 + w * K + i)
                torch.cuda.synchronize()
                ws.append(time.perf_counter() - t0)
        ws.sort()
        print(f"{wl} {bt}+{btu}  {name:26s} wall {1e6 * ws[2] / K:7.1f} us/step (best {1e6 * ws[0] / K:6.1f})", flush=True)
dist.destroy_process_group()
