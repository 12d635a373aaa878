"""The sharded step's collectives straight on RCCL (librccl through ctypes), for a host-bound rank.

Why.  Driven over torch.distributed a rank of the sharded step is bound by the HOST: a torch.distributed collective
costs 13-35 us of host time inside the step (Python -> c10d -> ProcessGroupNCCL: work object, watchdog enqueue, event
records, stream waits, the RCCL call) and the device idles 80 us of a 234-us step
(profiles/r06_dist_trace_b2_64.txt).  ``RcclComm`` owns an RCCL communicator of its own (``ncclCommInitRank``; the
unique id travels through the already initialised torch.distributed group, whatever its backend) and is used two ways:

  * ``native()``: as a ``cmlpl_collectives`` record (cmlpl_rccl_bind) -- ``DistTrainEngine.step`` then runs the WHOLE
    sharded step as one C call (cmlpl_dist_step, csrc/dist.hip), stages and collectives enqueued back to back: the
    product path (profiles/r06_dist_overhead_native.txt: host 53-58 us, wall 1.06-1.08 x the one-GPU engine);
  * with the interface of ``TorchDistComm`` (all_gather / reduce_scatter / all_reduce with ``async_op`` handles, as
    ``drive_step`` uses them; read_scalars / loss windows use these too): a synchronous collective is ONE ``nccl*``
    call on the current stream, an asynchronous one runs on a side stream between two hipEvents
    (``handle.wait()`` makes the current stream wait for the second).

``cmlpl_amd.distributed.pick_comm`` builds one for every engine on an RCCL process group, after a checked start-up
with a fallback to ``TorchDistComm``.  This pool has one GPU per box and RCCL refuses two ranks on one device, so this
class has RUN at world size 1 only (tests/test_gpu_rccl_comm.py: bit-identical to the torch.distributed path); the
staging around it is what the gloo / lockstep tests cover at W = 2, 4, 8."""
from __future__ import annotations

import ctypes as C
import os

import torch

NCCL_FLOAT32, NCCL_SUM = 7, 0


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]


def rccl_path() -> str:
    """the librccl.so this process uses: torch's own copy"""
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return path if os.path.exists(path) else "librccl.so"


def _load():
    rccl = C.CDLL(rccl_path())
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(vp), i32, _UniqueId, i32]
    rccl.ncclAllGather.argtypes = [vp, vp, sz, i32, vp, vp]
    rccl.ncclReduceScatter.argtypes = [vp, vp, sz, i32, i32, vp, vp]
    rccl.ncclAllReduce.argtypes = [vp, vp, sz, i32, i32, vp, vp]
    rccl.ncclCommDestroy.argtypes = [vp]
    rccl.ncclGetErrorString.restype = C.c_char_p
    hip = C.CDLL("libamdhip64.so")          # (the runtime torch has loaded: same handle)
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
    hip.hipEventRecord.argtypes = [vp, vp]
    hip.hipStreamWaitEvent.argtypes = [vp, vp, C.c_uint]
    hip.hipEventDestroy.argtypes = [vp]
    return rccl, hip


class RcclError(RuntimeError):
    pass


class RcclComm:
    class _Handle:
        def __init__(self, comm, ev):
            self.comm, self.ev = comm, ev

        def wait(self):
            c = self.comm
            c._hip_ok(c.hip.hipStreamWaitEvent(c._cur(), self.ev, 0), "hipStreamWaitEvent")

    def __init__(self, device, group=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RcclError("RcclComm: initialise torch.distributed first (the unique id travels through it)")
        self.device = torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.rccl, self.hip = _load()
        uid = _UniqueId()
        if self.rank == 0:
            self._ok(self.rccl.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [C.string_at(C.byref(uid), 128) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        C.memmove(C.byref(uid), box[0], 128)
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            self._ok(self.rccl.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.side = torch.cuda.Stream(device=self.device)
        self._side = C.c_void_p(self.side.cuda_stream)
        self._events = []
        for _ in range(4):                                   # two asynchronous collectives in flight at most: two events each
            ev = C.c_void_p()
            self._hip_ok(self.hip.hipEventCreateWithFlags(C.byref(ev), 2), "hipEventCreateWithFlags")   # hipEventDisableTiming
            self._events.append(ev)
        self._next = 0
        self._native = None

    # ---- plumbing
    def _ok(self, rc, what):
        if rc != 0:
            raise RcclError(f"{what}: {self.rccl.ncclGetErrorString(rc).decode()} ({rc})")

    @staticmethod
    def _hip_ok(rc, what):
        if rc != 0:
            raise RcclError(f"{what}: hipError_t {rc}")

    def _cur(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _issue(self, call, async_op):
        """call(stream) enqueues the collective; async: between two events on the side stream"""
        if not async_op:
            call(self._cur())
            return None
        e0, e1 = self._events[self._next], self._events[self._next + 1]
        self._next = (self._next + 2) % len(self._events)
        cur = self._cur()
        self._hip_ok(self.hip.hipEventRecord(e0, cur), "hipEventRecord")
        self._hip_ok(self.hip.hipStreamWaitEvent(self._side, e0, 0), "hipStreamWaitEvent")
        call(self._side)
        self._hip_ok(self.hip.hipEventRecord(e1, self._side), "hipEventRecord")
        return RcclComm._Handle(self, e1)

    @staticmethod
    def _f32(t):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise RcclError("RcclComm moves contiguous float32 device tensors")
        return t

    # ---- the three collectives of the sharded step
    def all_gather(self, out, inp, async_op=False):
        out, inp = self._f32(out), self._f32(inp)
        if out.numel() != self.world * inp.numel():
            raise RcclError("all_gather: out must hold world x inp elements")
        return self._issue(lambda st: self._ok(self.rccl.ncclAllGather(
            inp.data_ptr(), out.data_ptr(), inp.numel(), NCCL_FLOAT32, self.comm, st), "ncclAllGather"), async_op)

    def reduce_scatter(self, out, inp, async_op=False):
        out, inp = self._f32(out), self._f32(inp)
        if inp.numel() != self.world * out.numel():
            raise RcclError("reduce_scatter: inp must hold world x out elements")
        return self._issue(lambda st: self._ok(self.rccl.ncclReduceScatter(
            inp.data_ptr(), out.data_ptr(), out.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, st), "ncclReduceScatter"), async_op)

    def all_reduce(self, t, async_op=False):
        t = self._f32(t)
        return self._issue(lambda st: self._ok(self.rccl.ncclAllReduce(
            t.data_ptr(), t.data_ptr(), t.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, st), "ncclAllReduce"), async_op)

    def native(self, async_exchanges=None):
        """this communicator as a ``cmlpl_collectives`` record (cmlpl_rccl_bind): what ``cmlpl_dist_step`` -- the whole
        sharded step as ONE C call -- issues its four collectives through.  ``async_exchanges``: the embedding all-gather
        and the column-gradient reduce-scatter on the side stream, under the convolutions (a fork + join costs the step's
        stream ~10 us on this runtime whatever is forked: profiles/r06_event_hop_probe.txt -- it pays when the exchange
        itself takes longer than that on the wire -- at 8 GPUs the embedding all-gather should gain ~10 us, the reduce-scatter
        nothing: never measured, this pool has one GPU per box); False: all four in order on the step's stream, ONE stream for
        the communicator as torch.distributed itself uses it.  Default: in order; CMLPL_DIST_ASYNC=1 forks the two."""
        if self._native is None:
            from . import _lib
            rec = _lib.Collectives()
            with torch.cuda.device(self.device):
                _lib.check("cmlpl_rccl_bind", _lib.load().cmlpl_rccl_bind(rccl_path().encode(), self.comm, C.byref(rec)))
            self._native = rec
            self._native_inorder = _lib.Collectives.from_buffer_copy(rec)
            self._native_inorder.side_stream = None
        if async_exchanges is None:
            env = os.environ.get("CMLPL_DIST_ASYNC")
            async_exchanges = env is not None and env != "0"
        return self._native if async_exchanges else self._native_inorder

    def close(self):
        if getattr(self, "_native", None) is not None:
            from . import _lib
            torch.cuda.synchronize(self.device)
            _lib.load().cmlpl_rccl_unbind(C.byref(self._native))
            self._native = None
        if getattr(self, "comm", None):
            torch.cuda.synchronize(self.device)
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = None
            for ev in self._events:
                self.hip.hipEventDestroy(ev)
            self._events = []

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass
