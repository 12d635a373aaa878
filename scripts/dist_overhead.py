"""How long does the host take to ENQUEUE one data-parallel step vs how long the device takes to run it?"""
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
dist.destroy_process_group()
