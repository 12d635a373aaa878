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


def _cgroup_cpu():
    """(nr_throttled, throttled time in us) of this container's CPU controller (cgroup v2, else v1), or None"""
    for path, key_t in (("/sys/fs/cgroup/cpu.stat", "throttled_usec"), ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time")):
        try:
            d = dict(ln.split() for ln in open(path).read().splitlines() if len(ln.split()) == 2)
            t = int(d.get(key_t, 0))
            return int(d.get("nr_throttled", 0)), (t // 1000 if key_t == "throttled_time" else t)
        except OSError:
            continue
    return None


def _thread_cpu():
    """{tid: (name, CPU seconds)} of every thread of this process"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[t] = (name, (int(rest[11]) + int(rest[12])) / tick)
        except (OSError, ValueError):
            pass
    return out


def _sched():
    """this thread: (ns on a CPU, ns runnable but waiting for one, voluntary, involuntary context switches)"""
    import threading
    tid = threading.get_native_id()
    run, wait = (int(v) for v in open(f"/proc/self/task/{tid}/schedstat").read().split()[:2])
    st = dict(ln.split(":\t") for ln in open(f"/proc/self/task/{tid}/status").read().splitlines() if ":\t" in ln)
    return run, wait, int(st["voluntary_ctxt_switches"]), int(st["nonvoluntary_ctxt_switches"])


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
        h = None
        if key in self.copy:
            if inp is not None:
                out.view(-1).copy_(inp.view(-1))
        else:
            h = f()
        self.t[key] = self.t.get(key, 0.0) + time.perf_counter() - t0
        return h

    def all_gather(self, out, inp, async_op=False):
        return self._do("all_gather", lambda: TorchDistComm.all_gather(self, out, inp, async_op), out, inp)

    def reduce_scatter(self, out, inp, async_op=False):
        return self._do("reduce_scatter", lambda: TorchDistComm.reduce_scatter(self, out, inp, async_op), out, inp)

    def all_reduce(self, t, async_op=False):
        return self._do("all_reduce", lambda: TorchDistComm.all_reduce(self, t, async_op), t, None)


# A watcher thread: when the main thread has been inside one step for more than 5 ms, note what it is waiting in
# (/proc/<tid>/wchan = the kernel function it sleeps in, /proc/<tid>/syscall = the system call and its arguments,
# /proc/<tid>/stat field 3 = R running / S sleeping / D uninterruptible) -- once per stalled step.
import threading
MAIN_TID = threading.get_native_id()
STEP_T0 = [0.0]
WATCH = []


def _watch():
    seen = 0.0
    while True:
        time.sleep(0.001)
        t0 = STEP_T0[0]
        if t0 and t0 != seen and time.perf_counter() - t0 > 0.005:
            seen = t0
            rec = {}
            for f in ("wchan", "syscall", "stat"):
                try:
                    v = open(f"/proc/self/task/{MAIN_TID}/{f}").read().strip()
                    rec[f] = v.split()[2] if f == "stat" else v[:120]
                except OSError as e:
                    rec[f] = f"({e.errno})"
            rec["after_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
            WATCH.append(rec)


threading.Thread(target=_watch, daemon=True).start()


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
    WATCH.clear()
    tc0, tw0 = _thread_cpu(), time.perf_counter()
    cg0, worst = _cgroup_cpu(), (0.0, None)
    t0 = time.perf_counter()
    for i in range(K):
        comm.site = 0
        s0 = _sched()
        c0 = time.thread_time()
        ts = time.perf_counter()
        STEP_T0[0] = ts
        eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, 20 + i)
        STEP_T0[0] = 0.0
        dt = time.perf_counter() - ts
        per.append(dt)
        if dt > worst[0]:
            s1 = _sched()
            worst = (dt, dict(step=i, thread_cpu_ms=round(1e3 * (time.thread_time() - c0), 2), on_cpu_ms=round((s1[0] - s0[0]) / 1e6, 2),
                              runnable_waiting_ms=round((s1[1] - s0[1]) / 1e6, 2), voluntary_switches=s1[2] - s0[2],
                              involuntary_switches=s1[3] - s0[3]))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per.sort()
    tag = f"{wl} {bt}+{btu}" + (f" copy={sorted(copy)}" if copy else "")
    print(f"{tag}: host {1e6 * (t1 - t0) / K:7.1f} us/step  wall {1e6 * (t2 - t0) / K:7.1f}  per-step host median {1e6 * per[K // 2]:.1f} "
          f"p90 {1e6 * per[int(K * 0.9)]:.1f} max {1e6 * per[-1]:.1f}")
    print("    stages      " + "  ".join(f"{k} {1e6 * v / K:.1f}" for k, v in st.items()))
    print("    collectives " + "  ".join(f"{k} {1e6 * v / K:.1f}" for k, v in comm.t.items()))
    cg1 = _cgroup_cpu()
    print(f"    slowest step {1e3 * worst[0]:.2f} ms: {worst[1]}   cgroup throttling during the run: "
          + (f"{cg1[0] - cg0[0]} periods, {(cg1[1] - cg0[1]) / 1e3:.1f} ms" if cg0 and cg1 else "n/a")
          + f"   threads {len(os.listdir('/proc/self/task'))}")
    tc1, tw = _thread_cpu(), time.perf_counter() - tw0
    burn = sorted(((c - tc0.get(t, (n, 0.0))[1], n, t) for t, (n, c) in tc1.items()), reverse=True)
    tot = sum(b[0] for b in burn)
    print(f"    CPU burnt by this process's threads over the {1e3 * tw:.0f} ms of the run: {1e3 * tot:.0f} ms = {tot / tw:.1f} cores; top: "
          + ", ".join(f"{n}[{t}] {1e3 * c:.0f} ms" for c, n, t in burn[:6]))
    if WATCH:
        print(f"    watcher (main thread inside a step for > 5 ms): {WATCH}")
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
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError:
    try:
        print("cgroup cfs quota / period (us):", open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read().strip(), "/",
              open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
    except OSError:
        print("cgroup cpu limits: not readable")
print("online CPUs:", os.cpu_count(), " affinity:", len(os.sched_getaffinity(0)))
dist.destroy_process_group()
