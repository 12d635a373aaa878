"""What does the staged data-parallel step cost before any wire time, and what do its four torch.distributed calls cost
the HOST?  At world size 1 on one GPU (RCCL initialised):
  TrainEngine                      one C call per step (the single-GPU engine)
  DistTrainEngine, aliased         seven stage calls, the one-rank collectives aliased away (NoOpComm)
  DistTrainEngine, RCCL direct     the same with the four collectives as direct librccl calls (cmlpl_amd/rccl_comm.py)
  DistTrainEngine, real calls      the same with the four REAL collectives per step (all_gather_into_tensor x 2,
                                   reduce_scatter_tensor, all_reduce through TorchDistComm; separate send / receive buffers)
host enqueue = wall time to ENQUEUE a step; wall = steps including the final synchronisation; plus the host time of each
collective call alone.  Sizes: the headline batch and the per-rank shards of BASELINE configs[2] / configs[4] at 8 GPUs.
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
from cmlpl_amd.distributed import DistTrainEngine, TorchDistComm
from bench import synth, WORKLOADS
from cmlpl_amd.rccl_comm import RcclComm
RCCL = RcclComm(dev)

K = 200
for wl, bt, btu in (("B2", 128, 128), ("B2", 64, 64), ("B5", 8, 64)):
    shape = WORKLOADS[wl]
    b = synth(shape, bt, btu, 1, dev)
    print(f"{wl} {bt}+{btu} rows")
    walls = []
    for name, make in (("TrainEngine", lambda: TrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088)),
                       ("DistTrainEngine aliased", lambda: DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088)),
                       ("DistTrainEngine RCCL direct", lambda: DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088,
                                                                                comm=RCCL, alias_single=False)),
                       ("DistTrainEngine real calls", lambda: DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088,
                                                                               comm=TorchDistComm(), alias_single=False))):
        eng = make()
        eng.init_params_default(1088)
        for i in range(20):
            eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, 20 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        walls.append(t2 - t0)
        print(f"  {name:28s} host enqueue {1e6 * (t1 - t0) / K:7.1f} us/step   wall {1e6 * (t2 - t0) / K:7.1f} us/step", flush=True)
    print(f"  real calls / TrainEngine wall: {walls[3] / walls[0]:.3f}   RCCL direct / TrainEngine wall: {walls[2] / walls[0]:.3f}")
    # ... and the same engine replaying its seven stage graphs (DistStepGraph), the four real calls eager between them
    li, ui = torch.arange(bt, device=dev), torch.arange(btu, device=dev)
    gr = eng.capture(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], li, ui, bt, btu, capacity=K + 20)
    gr.program([(1, 300 + i, 0, 0) for i in range(K + 20)])
    for i in range(20):
        gr.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        gr.launch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"  {'DistStepGraph + real calls':28s} host enqueue {1e6 * (t1 - t0) / K:7.1f} us/step   wall {1e6 * (t2 - t0) / K:7.1f} us/step"
          f"   (host / device = {(t1 - t0) / (t2 - t0):.2f})", flush=True)
    gr.close()
    # the four calls alone (host time of the call, device idle)
    comm = TorchDistComm()
    for kind, out, inp, _key in [x for st in eng.STAGES for x in eng.exchange_after(st)]:
        f = {"all_gather": lambda: comm.all_gather(out, inp), "reduce_scatter": lambda: comm.reduce_scatter(out, inp),
             "all_reduce": lambda: comm.all_reduce(out)}[kind]
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            f()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"    {kind:15s} {out.numel() * 4 / 1e6:6.2f} MB: host {1e6 * (t1 - t0) / K:6.1f} us/call, back to back {1e6 * (t2 - t0) / K:6.1f} us/call")
dist.destroy_process_group()
