"""What does the staged data-parallel step cost before any wire time?  DistTrainEngine at world size 1 (RCCL initialised,
the four collectives of a step being identities) against TrainEngine on the same batch: host enqueue time and wall time
per step.  Round 1: +24 % (a re-ordering copy of the gathered buffer, a label copy, RCCL's identity copies).  Now the
loss kernels read the gathered blocks in place, the labels are written into the exchange buffer by the forward, and
one-rank collectives alias their buffers:
    python scripts/dist_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29514")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from cmlpl_amd import NetShape, HyperParams, TrainEngine
from cmlpl_amd.distributed import DistTrainEngine
from bench import synth, WORKLOADS
shape = WORKLOADS["B2"]
for cls in (TrainEngine, DistTrainEngine):
    eng = cls(NetShape(*shape), 128, 128, HyperParams(), device=dev, seed=1088)
    eng.init_params_default(1088)
    b = synth(shape, 128, 128, 1, dev)
    for i in range(20):
        eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, i)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K):
        eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, 20 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{cls.__name__:16s} host enqueue {1e6 * (t1 - t0) / K:7.1f} us/step   wall {1e6 * (t2 - t0) / K:7.1f} us/step", flush=True)
    walls = globals().setdefault("walls", [])
    walls.append(t2 - t0)
print(f"DistTrainEngine / TrainEngine wall time at world size 1: {walls[1] / walls[0]:.3f}")
dist.destroy_process_group()
