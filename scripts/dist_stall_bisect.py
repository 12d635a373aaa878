"""Where does the host go in the eager sharded step with REAL collectives?  (`dist_overhead.py` showed
418-431 us of host time per step at B2 64 + 64 against 103-107 at the two other shard sizes, in two records.)

Per shard size, one DistTrainEngine with the four real torch.distributed calls at world size 1; host time of EVERY call
of a step (five stage calls, four collectives) accumulated separately, then the same with the configs in another order
and with each collective in turn replaced by a plain copy -- so that a stall can be pinned on one call, one buffer or
the order of the runs.
    python scripts/dist_stall_bisect.py [order]      order: e.g. "B2:64:64,B2:128:128,B5:8:64" (default: the r05 order)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29516")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from cmlpl_amd import NetShape, HyperParams
from cmlpl_amd.distributed import DistTrainEngine, TorchDistComm
from bench import synth, WORKLOADS

K = 200
import gc
GC_LOG = []          # (generation, seconds) of every collection of the cyclic garbage collector
_gc_t0 = [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t0[0] = time.perf_counter()
    else:
        GC_LOG.append((info["generation"], time.perf_counter() - _gc_t0[0]))


gc.callbacks.append(_gc_cb)


class TimedComm(TorchDistComm):
    """real collectives, host time per call site; `copy` names call sites served by a plain copy instead"""
    def __init__(self, copy=()):
        super().__init__()
        self.t = {}
        self.copy = set(copy)
        self.site = 0

    def _do(self, name, f, out, inp):
        key = f"{self.site}:{name}"
        self.site += 1
        t0 = time.perf_counter()
        if key in self.copy:
            if inp is not None:
                out.view(-1).copy_(inp.view(-1))
        else:
            f()
        self.t[key] = self.t.get(key, 0.0) + time.perf_counter() - t0

    def all_gather(self, out, inp): self._do("all_gather", lambda: TorchDistComm.all_gather(self, out, inp), out, inp)
    def reduce_scatter(self, out, inp): self._do("reduce_scatter", lambda: TorchDistComm.reduce_scatter(self, out, inp), out, inp)
    def all_reduce(self, t): self._do("all_reduce", lambda: TorchDistComm.all_reduce(self, t), t, None)


def run(wl, bt, btu, copy=()):
    shape = WORKLOADS[wl]
    b = synth(shape, bt, btu, 1, dev)
    comm = TimedComm(copy)
    eng = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088, comm=comm, alias_single=False)
    eng.init_params_default(1088)
    st = {}
    for name in eng.STAGES:                       # host time of each stage call
        f = getattr(eng, "stage_" + name)
        def wrap(*a, _f=f, _n=name, **k):
            t0 = time.perf_counter()
            r = _f(*a, **k)
            st[_n] = st.get(_n, 0.0) + time.perf_counter() - t0
            return r
        setattr(eng, "stage_" + name, wrap)
    for i in range(20):
        comm.site = 0
        eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, i)
    torch.cuda.synchronize()
    st.clear(); comm.t.clear()
    per = []
    GC_LOG.clear()
    t0 = time.perf_counter()
    for i in range(K):
        comm.site = 0
        ts = time.perf_counter()
        eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, 20 + i)
        per.append(time.perf_counter() - ts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per.sort()
    tag = f"{wl} {bt}+{btu}" + (f" copy={sorted(copy)}" if copy else "")
    print(f"{tag}: host {1e6 * (t1 - t0) / K:7.1f} us/step  wall {1e6 * (t2 - t0) / K:7.1f}  per-step host median {1e6 * per[K // 2]:.1f} "
          f"p90 {1e6 * per[int(K * 0.9)]:.1f} max {1e6 * per[-1]:.1f}")
    print("    stages      " + "  ".join(f"{k} {1e6 * v / K:.1f}" for k, v in st.items()))
    print("    collectives " + "  ".join(f"{k} {1e6 * v / K:.1f}" for k, v in comm.t.items()))
    print("    garbage collections during the timed steps: " +
          (", ".join(f"gen{g} {1e3 * t:.1f} ms" for g, t in GC_LOG) or "none") + f"   (gc enabled: {gc.isenabled()})", flush=True)
    del eng


order = sys.argv[1] if len(sys.argv) > 1 else "B2:128:128,B2:64:64,B5:8:64,B2:64:64"
for item in order.split(","):
    wl, bt, btu = item.split(":")
    run(wl, int(bt), int(btu))
print("-- B2 64+64 with one collective at a time replaced by a copy")
for c in ("0:all_gather", "1:all_gather", "2:reduce_scatter", "3:all_reduce"):
    run("B2", 64, 64, copy=(c,))
run("B2", 64, 64, copy=("0:all_gather", "1:all_gather", "2:reduce_scatter", "3:all_reduce"))
print("-- the same sizes with the collector frozen and switched off (gc.freeze(); gc.disable())")
gc.collect(); gc.freeze(); gc.disable()
for item in order.split(","):
    wl, bt, btu = item.split(":")
    run(wl, int(bt), int(btu))
dist.destroy_process_group()
