"""What does the staged data-parallel step cost before any wire time, and what do its four collectives cost the HOST?
At world size 1 on one GPU (RCCL initialised):
  TrainEngine                      one C call per step (the single-GPU engine)
  sharded, one call, aliased       cmlpl_dist_step, the one-rank collectives aliased away (NoOpComm)
  sharded, one call, RCCL          cmlpl_dist_step with the four REAL collectives issued from C (cmlpl_rccl_bind)
  sharded, Python, aliased         seven stage calls from Python (drive_step), collectives aliased away
  sharded, Python, RCCL direct     drive_step + the collectives as direct librccl calls through ctypes (RcclComm)
  sharded, Python, torch.dist      drive_step + all_gather_into_tensor x 2, reduce_scatter_tensor, all_reduce (TorchDistComm)
host enqueue = wall time to ENQUEUE a step; wall = steps including the final synchronisation; plus the host time of each
torch.distributed call alone.  Sizes: the headline batch and the per-rank shards of BASELINE configs[2] / configs[4] at 8 GPUs.
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
from cmlpl_amd.distributed import DistTrainEngine, NoOpComm, TorchDistComm
from bench import synth, WORKLOADS
from cmlpl_amd.rccl_comm import RcclComm
RCCL = RcclComm(dev)

K = 100
for wl, bt, btu in (("B2", 128, 128), ("B2", 64, 64), ("B5", 8, 64)):
    shape = WORKLOADS[wl]
    b = synth(shape, bt, btu, 1, dev)
    print(f"{wl} {bt}+{btu} rows")
    walls = []
    def sharded(comm, native):
        def make():
            e = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088, comm=comm, alias_single=False)
            e.native_step = native
            return e
        return make
    for name, make in (("TrainEngine", lambda: TrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088)),
                       ("sharded, one call, aliased", sharded(NoOpComm(), True)),
                       ("sharded, one call, RCCL", sharded(RCCL, True)),
                       ("sharded, Python, aliased", sharded(NoOpComm(), False)),
                       ("sharded, Python, RCCL direct", sharded(RCCL, False)),
                       ("sharded, Python, torch.dist", sharded(TorchDistComm(), False))):
        eng = make()
        eng.init_params_default(1088)
        for i in range(20):
            eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, i)
        torch.cuda.synchronize()
        # five windows; the median window counts (a window the container's CPU quota froze for tens of milliseconds --
        # profiles/r06_stall_rootcause.txt -- shows up in "worst" only)
        hs, ws = [], []
        for w in range(5):
            t0 = time.perf_counter()
            for i in range(K):
                eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, 20 + w * K + i)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            hs.append(t1 - t0); ws.append(t2 - t0)
        hs.sort(); ws.sort()
        walls.append(ws[2])
        print(f"  {name:30s} host enqueue {1e6 * hs[2] / K:7.1f} us/step   wall {1e6 * ws[2] / K:7.1f} us/step   "
              f"(best {1e6 * ws[0] / K:6.1f}, worst {1e6 * ws[4] / K:6.1f})", flush=True)
    print(f"  wall / TrainEngine wall: one call + RCCL {walls[2] / walls[0]:.3f}   Python + RCCL direct {walls[4] / walls[0]:.3f}   Python + torch.distributed {walls[5] / walls[0]:.3f}")
    # ... and the same engine replaying its seven stage graphs (DistStepGraph), the four real calls eager between them
    li, ui = torch.arange(bt, device=dev), torch.arange(btu, device=dev)
    gr = eng.capture(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], li, ui, bt, btu, capacity=2 * K + 20)
    gr.program([(1, 300 + i, 0, 0) for i in range(2 * K + 20)])
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
