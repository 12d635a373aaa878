"""Sample-sharded data parallelism for the CMLPL step, one process per GPU.

The reference is single-process; this layer is the build's own design for one xGMI node:
both networks and both memory banks are replicated, every rank takes bt/W labelled + btu/W unlabelled
rows, and a step exchanges exactly what crosses samples -- staged by what depends on what in
tools/models.py:130-152, so that only two small collectives sit on the critical path:

  1. spectral    augmented spectra -> feat_spe + ReLU -> L2 norm: the EMBEDDINGS (they depend on nothing the
                 convolutions compute, models.py:142-146)
       ALL-GATHER (async) [feat | labels] of every rank  -> the keys of the contrastive terms, the bank rows
  2. spatial     conv0 .. conv2 + head -> this rank's logits            ... the gather runs under it
  3. phase 1     (waits for the gather) similarity tiles, per-row CE / softmax / smoothing / masks / mutual loss
       ALL-GATHER the smoothed + un-smoothed probabilities [4][btu/W][K] (18 KB: the pseudo-label graph
                  Q0 = p_s.p_w^T and the bank write need every row's)           <- exposed
  4. phase 2     local rows x global keys contrastive loss, identical bank write on every rank
       REDUCE-SCATTER (async) the column-side gradient d fU_w [btu][1024]  (backward of gathering the keys)
  5. backward data     classifier / conv2 / conv1 data gradients, conv0 + 3x3 weight-gradient partials: needs
                       dlogits ONLY (models.py:144-150)                 ... the reduce-scatter runs under it
  6. backward weights  (waits for the reduce-scatter) dy takes its feature-gradient share, partials are
                       reduced, classifier / feat_spe weight gradients
       ALL-REDUCE(sum) one flat gradient bucket (both networks, 2 x 207,881 floats for PaviaU)   <- exposed
                  losses are normalised by GLOBAL counts inside the kernels, so the sum IS the global gradient
  7. update      Adam on every rank (replicated, deterministic)

Four collectives per step (the logits are never gathered: only this rank's rows are read, the bank takes
the other rows' probabilities from collective 2), all NCCL(=RCCL)-over-xGMI through torch.distributed;
message sizes are 18 KB - 4 MB, i.e. latency-bound on 7 x 153 GB/s links, hence one packed buffer per
exchange rather than one collective per tensor.  Through round 5 all four ran synchronously between the
stages; the two large ones now overlap the convolutions.  W-rank results equal the 1-rank results on the
same global batch up to fp32 summation order (tests/test_distributed_*.py).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
from .config import FEAT_DIM, HyperParams, NetShape
from .engine import SCALAR_NAMES, TrainEngine, _chk_f32


class DistStartupError(RuntimeError):
    """the job's ranks cannot run as one process per GPU (raised before the first training step)"""


def _fail_after(seconds: float, what: str):
    """Watchdog for a start-up phase that may hang (rendezvous, RCCL communicator creation, first collective): if the
    returned timer is not cancelled in time, say what was being waited for and end THIS process with code 3 -- a hung
    rank would otherwise hang the whole job (the launcher stops the others when one exits non-zero)."""
    import os
    import sys
    import threading

    def boom():
        sys.stderr.write(f"cmlpl_amd.distributed: rank {os.environ.get('RANK', '?')} gave up after {seconds:.0f} s "
                         f"waiting for {what}\n")
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(seconds, boom)
    t.daemon = True
    t.start()
    return t


def init_distributed(backend: str = "nccl", device: Optional[torch.device] = None, timeout_s: float = 300.0,
                     one_gpu: bool = False):
    """Create the process group of a one-process-per-GPU job and CHECK it before anything else runs on it.

    Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run or cmlpl_amd.launch set them).  Raises
    DistStartupError -- with a message that says what to fix -- when the node cannot give every local rank its own
    GPU, or when the first collectives return wrong data; exits with code 3 (message on stderr) when rendezvous or the
    first collective does not finish within `timeout_s`, instead of hanging.  Checks, in order:
      1. backend "nccl" (= RCCL on ROCm): LOCAL_WORLD_SIZE visible devices at least, LOCAL_RANK < device count;
      2. init_process_group under the watchdog (RCCL: bound to `device`, its timeout set to `timeout_s`);
      3. an all-gather of (rank, device index, PCI bus id) -- every local rank must sit on a DIFFERENT device
         (RCCL refuses two ranks on one GPU late and cryptically) -- and an all-reduce of rank + 1 whose sum must be
         W (W + 1) / 2 on every rank.
    `one_gpu`: rehearsal of the multi-process path on a one-GPU box (gloo over device buffers): check 1 / the
    distinct-device part of 3 are skipped.  Returns torch.distributed."""
    import datetime
    import os
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if backend == "nccl" and not one_gpu:
        ndev = torch.cuda.device_count()
        if ndev < local_world or local_rank >= ndev:
            raise DistStartupError(
                f"rank {rank}: {local_world} ranks on this node but {ndev} visible GPU(s) (LOCAL_RANK {local_rank}): "
                "one process per GPU needs one device per local rank -- check HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES "
                "and --nproc-per-node")
    guard = _fail_after(timeout_s, "the process-group rendezvous (MASTER_ADDR/MASTER_PORT reachable? all ranks started?)")
    try:
        kw = dict(timeout=datetime.timedelta(seconds=timeout_s))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)
    except Exception as e:
        from .launch import EXIT_PORT_TAKEN, port_taken
        if port_taken(e):          # lost the port between the launcher's probe and this bind: the launcher starts over
            import sys
            sys.stderr.write(f"cmlpl_amd.distributed: rank {rank}: rendezvous port already in use ({e})\n")
            sys.stderr.flush()
            os._exit(EXIT_PORT_TAKEN)
        raise
    finally:
        guard.cancel()
    guard = _fail_after(timeout_s, f"the first {backend} collective (RCCL: HSA_ENABLE_IPC_MODE_LEGACY=0 set? xGMI peers visible?)")
    try:
        on = device if (backend == "nccl" and device is not None) else torch.device("cpu")
        idx = -1 if device is None or device.index is None else int(device.index)
        bus = 0
        if device is not None and device.type == "cuda":
            try:
                bus = int(torch.cuda.get_device_properties(device).pci_bus_id)
            except Exception:
                bus = idx
        mine = torch.tensor([rank, idx, bus, local_rank], dtype=torch.int64, device=on)
        allr = torch.zeros(world * 4, dtype=torch.int64, device=on)
        dist.all_gather_into_tensor(allr, mine)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=on)
        dist.all_reduce(t)
        if on.type == "cuda":
            torch.cuda.synchronize(on)
        rows = allr.view(world, 4).cpu().tolist()
    finally:
        guard.cancel()
    if [r[0] for r in rows] != list(range(world)):
        raise DistStartupError(f"rank {rank}: all-gather returned ranks {[r[0] for r in rows]}")
    if abs(float(t.item()) - world * (world + 1) / 2) > 1e-9:
        raise DistStartupError(f"rank {rank}: all-reduce of rank+1 gave {float(t.item())}, expected {world * (world + 1) / 2}")
    if backend == "nccl" and not one_gpu and world == local_world:
        seen = {}
        for r, dev_i, bus_i, lr in rows:
            key = (dev_i, bus_i)
            if key in seen:
                raise DistStartupError(f"ranks {seen[key]} and {r} share GPU {dev_i} (PCI bus {bus_i}): give every "
                                       "local rank its own device (device = cuda:LOCAL_RANK)")
            seen[key] = r
    return dist


def pick_comm(device, timeout_s: float = 120.0):
    """The communicator of a sharded engine on an initialised process group.  Backend nccl: the collectives straight
    on RCCL (``RcclComm``: the whole step then runs as ONE C call, cmlpl_dist_step) -- after a checked start-up: every
    rank builds it under a watchdog and runs its three collectives on random data against torch.distributed's; all
    ranks must agree that they match (an all-reduce(MIN) through torch.distributed), otherwise every rank falls back to
    ``TorchDistComm`` (the step driven stage by stage from Python) and says so on stderr.  CMLPL_DIST_COMM=torch skips
    the attempt, =rccl makes a failed check an error.  Any other backend (gloo rehearsals): ``TorchDistComm``."""
    import sys
    import torch.distributed as dist
    want = os.environ.get("CMLPL_DIST_COMM", "auto")
    if want == "torch" or dist.get_backend() != "nccl":
        return TorchDistComm()
    W, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(device)
    rc, ok, why = None, 1.0, ""
    guard = _fail_after(timeout_s, "the start-up of the direct RCCL communicator (CMLPL_DIST_COMM=torch skips it)")
    try:
        try:
            from .rccl_comm import RcclComm
            rc = RcclComm(dev)
            g = torch.Generator(device="cpu").manual_seed(77 + rank)
            a = torch.randn(1024 * W, generator=g).to(dev)
            o1, o2 = torch.empty(1024 * W * W, device=dev), torch.empty(1024 * W * W, device=dev)
            rc.all_gather(o1, a); dist.all_gather_into_tensor(o2, a)
            s1, s2 = torch.empty(1024, device=dev), torch.empty(1024, device=dev)
            h = rc.reduce_scatter(s1, a, async_op=True); h.wait(); dist.reduce_scatter_tensor(s2, a.clone())
            r1, r2 = a.clone(), a.clone()
            rc.all_reduce(r1); dist.all_reduce(r2)
            torch.cuda.synchronize(dev)
            if not (torch.equal(o1, o2) and torch.allclose(s1, s2, rtol=1e-5, atol=1e-5) and torch.allclose(r1, r2, rtol=1e-5, atol=1e-5)):
                ok, why = 0.0, "its collectives differ from torch.distributed's"
        except Exception as e:       # noqa: BLE001  (a missing symbol, an RCCL error: the fallback is the answer)
            ok, why = 0.0, f"{type(e).__name__}: {e}"
        t = torch.tensor([ok], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        all_ok = float(t.item()) == 1.0
    finally:
        guard.cancel()
    if all_ok:
        return rc
    if want == "rccl":
        raise DistStartupError(f"rank {rank}: CMLPL_DIST_COMM=rccl but the direct RCCL communicator failed its check ({why or 'on another rank'})")
    sys.stderr.write(f"cmlpl_amd.distributed: rank {rank}: direct RCCL communicator not used ({why or 'failed on another rank'}); "
                     "collectives go through torch.distributed\n")
    if rc is not None:
        try:
            rc.close()
        except Exception:            # noqa: BLE001
            pass
    return TorchDistComm()


class TorchDistComm:
    """torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" on CPU).

    Stream hand-off.  The HIP kernels of a stage are launched on torch's CURRENT stream of the engine's device.  A
    torch.distributed collective called with async_op=False on a CUDA tensor is enqueued on ProcessGroupNCCL's own
    stream behind an event recorded on the current stream, and the current stream is made to wait for the collective's
    end event before the call returns: producer kernels -> collective -> consumer kernels are ordered on the device
    without a host synchronisation.  `debug=True` (CMLPL_DIST_DEBUG=1) asserts it after every collective: the data
    that must have arrived is checked on the host (own block of an all-gather bit-equal to the input; float64 sums
    before / after a reduce-scatter and an all-reduce agree across ranks) -- slow, for bring-up on a new node."""

    def __init__(self, group=None, debug: Optional[bool] = None):
        import os
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.debug = bool(int(os.environ.get("CMLPL_DIST_DEBUG", "0"))) if debug is None else bool(debug)

    class _Handle:
        """an asynchronous collective in flight: wait() orders the CURRENT stream behind it (no host block on RCCL)
        and runs the debug check of the finished exchange"""
        def __init__(self, work, after=None):
            self.work, self.after = work, after

        def wait(self):
            if self.work is not None:
                self.work.wait()
            if self.after is not None:
                self.after()

    def _finish(self, work, async_op, after):
        if async_op:
            return TorchDistComm._Handle(work, after)
        if after is not None:
            after()
        return None

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        work = self.dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group, async_op=async_op)
        after = None
        if self.debug:
            def after():
                n = inp.numel()
                got = out.view(-1)[self.rank * n:(self.rank + 1) * n]
                if not torch.equal(got.cpu(), inp.view(-1).cpu()):
                    raise RuntimeError(f"rank {self.rank}: all_gather did not deliver this rank's own block (stream order?)")
        return self._finish(work, async_op, after)

    def _check(self, name, before, after):
        """debug: [sum, sum of |.|] over all ranks before and after a reduction must agree to fp32 summation error"""
        self.dist.all_reduce(before, group=self.group)
        self.dist.all_reduce(after, group=self.group)
        b, a = before.tolist(), after.tolist()
        if not abs(a[0] - b[0]) <= 1e-5 * b[1] + 1e-6:
            raise RuntimeError(f"rank {self.rank}: {name} checksum {a[0]} != {b[0]} (stream order?)")

    @staticmethod
    def _sums(t):
        d = t.detach().double()
        return torch.stack([d.sum(), d.abs().sum()])

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        before = self._sums(inp) if self.debug else None
        work = self.dist.reduce_scatter_tensor(out.view(-1), inp.view(-1), op=self.dist.ReduceOp.SUM, group=self.group,
                                               async_op=async_op)
        after = None
        if self.debug:      # sum over ranks of the inputs == sum over ranks of the scattered outputs
            def after():
                aft = self._sums(out)
                aft[1] = before[1]
                self._check("reduce_scatter", before, aft)
        return self._finish(work, async_op, after)

    def all_reduce(self, t: torch.Tensor, async_op: bool = False):
        before = self._sums(t) if self.debug else None
        work = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        after = None
        if self.debug:      # every rank now holds the sum: W x (sum over ranks of the inputs) after the second reduction
            def after():
                aft = self._sums(t) / self.world
                aft[1] = before[1]
                self._check("all_reduce", before, aft)
        return self._finish(work, async_op, after)


class SingleComm:
    """world size 1: every collective is a copy."""
    world, rank = 1, 0

    def all_gather(self, out, inp, async_op=False): out.view(-1).copy_(inp.view(-1))
    def reduce_scatter(self, out, inp, async_op=False): out.view(-1).copy_(inp.view(-1))
    def all_reduce(self, t, async_op=False): pass


class NoOpComm:
    """world size 1: a one-rank collective is the identity, and the engine aliases the buffers on both sides of it
    (see DistTrainEngine._bind), so nothing is launched."""
    world, rank = 1, 0

    def all_gather(self, out, inp, async_op=False): assert out.data_ptr() == inp.data_ptr()
    def reduce_scatter(self, out, inp, async_op=False): assert out.data_ptr() == inp.data_ptr()
    def all_reduce(self, t, async_op=False): pass


def issue_exchange(comm, spec):
    """one exchange of a stage: spec = (kind, out, inp, key); key None = synchronous, else the handle (or None, when the
    backend finished it on the spot) is returned for the stage that waits for it"""
    kind, out, inp, key = spec
    a = key is not None
    if kind == "all_gather":
        return comm.all_gather(out, inp, async_op=a)
    if kind == "reduce_scatter":
        return comm.reduce_scatter(out, inp, async_op=a)
    if kind == "all_reduce":
        return comm.all_reduce(out, async_op=a)
    raise ValueError(kind)


def drive_step(engine, comm, *fwd_args, **fwd_kw) -> None:
    """Run the stages of one sharded step with the collectives that follow them.  ``engine`` provides STAGES,
    stage_<name>(), exchange_after(name) -> [(kind, out, inp, key)] and waits_before(name) -> [key]
    (DistTrainEngine, or the CPU stand-in used by the gloo tests).  An exchange with a key is issued asynchronously
    right behind the stage that produced its input and waited for in front of the first stage that reads its output:
    the embedding all-gather overlaps the convolution forward, the column-gradient reduce-scatter the convolution
    backward."""
    pending = {}
    for i, stage in enumerate(engine.STAGES):
        for key in engine.waits_before(stage):
            h = pending.pop(key)
            if h is not None:
                h.wait()
        if i == 0:
            getattr(engine, "stage_" + stage)(*fwd_args, **fwd_kw)
        else:
            getattr(engine, "stage_" + stage)()
        for spec in engine.exchange_after(stage):
            h = issue_exchange(comm, spec)
            if spec[3] is not None:
                pending[spec[3]] = h
    assert not pending, f"exchanges never waited for: {sorted(pending)}"


class DistTrainEngine(TrainEngine):
    """TrainEngine whose step is sharded by sample over ``comm.world`` ranks.  Batch sizes given to
    the constructor are PER RANK (the largest shard a step may bring); banks are sized from the global
    labelled batch (train.py:138).  A step may bring fewer rows (the last short batch of an epoch,
    train.py:134 keeps it) as long as EVERY rank brings the same number: shards are equal by construction,
    which is what makes the sum of the ranks' shares the global mean."""

    STAGES = ("spectral", "spatial", "phase1", "phase2", "backward_data", "backward_weights", "update")

    def __init__(self, shape: NetShape, labeled_batch_size: int, unlabeled_batch_size: int,
                 hp: Optional[HyperParams] = None, device="cuda:0", seed: int = 1088, comm=None, hist_rows: int = 1,
                 alias_single: bool = True):
        """``alias_single=False`` keeps the REAL collectives at world size 1 (separate send / receive buffers, four
        torch.distributed calls per step): what scripts/dist_overhead.py measures the host cost of the calls with."""
        if comm is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                comm = NoOpComm()
            elif dist.get_world_size() == 1 and alias_single:
                comm = NoOpComm()
            else:
                comm = pick_comm(device)
        if comm.world == 1 and alias_single and not isinstance(comm, NoOpComm) and hasattr(comm, "all_gather"):
            comm = NoOpComm()      # identity collectives: alias instead of launching RCCL copies
        self.comm = comm
        self.native_step = os.environ.get("CMLPL_DIST_NATIVE", "1") != "0"
        self.async_exchanges = None      # the one-call step: None = the communicator's default (RcclComm.native)
        W = self.world = comm.world
        self.rank = comm.rank
        super().__init__(shape, labeled_batch_size, unlabeled_batch_size, hp, device, seed,
                         bank_labeled=labeled_batch_size * W, hist_rows=hist_rows)
        bt_l, btu_l, K = self.bt_max, self.btu_max, shape.K
        n_l = bt_l + btu_l
        dev = self.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        # ONE gradient bucket: the live tensors of both networks back to back ([2][live], the kernels take the
        # per-network stride as an argument) -> one all-reduce of 2 x 207,881 floats instead of two
        self.live = int(self.layout.param_live)
        self.grads = z(2, self.live)
        # flat max-size buffers; a step's tensors are views of their heads (see _bind)
        self._pack = z(2 * n_l * FEAT_DIM + bt_l)              # this rank's block of the exchange buffer: [feat | labels]
        self._recv = z(W * self._pack.numel())
        self._logits_loc = z(2 * n_l * K)                      # this rank's logits: never exchanged by the step
        self._recv_z = z(W * 2 * n_l * K)                      # outputs(): the logits gathered on request
        self._logits_g, self._feat_g = z(2 * n_l * W * K), z(2 * n_l * W * FEAT_DIM)
        self._labels_g = torch.zeros(bt_l * W, dtype=torch.int64, device=dev)
        self._probs_l, self._probs_g = z(4 * btu_l * K), z(W * 4 * btu_l * K)
        self._dlogits_l, self._dfeat_l = z(2 * n_l * K), z(2 * n_l * FEAT_DIM)
        self._dfw_part = z(btu_l * W * FEAT_DIM)
        lw = self.lib.cmlpl_loss_workspace_bytes(
            C.byref(self.cshape), C.byref(_lib.Shard(bt_l * W, btu_l * W, self.rank * bt_l, bt_l, self.rank * btu_l, btu_l)),
            self.Q)
        if lw == 0:
            raise _lib.CmlplError("cmlpl_loss_workspace_bytes", -2)
        self.loss_ws = torch.empty(lw, dtype=torch.uint8, device=dev)
        self._bound = None
        self._bind(bt_l, btu_l)
        self._ctx = None

    def _bind(self, bt_l: int, btu_l: int) -> None:
        """Views for a step of bt_l + btu_l rows on every rank."""
        if self._bound == (bt_l, btu_l):
            return
        W, K, s = self.world, self.shape.K, self.shape
        n_l = bt_l + btu_l
        self.bt_l, self.btu_l = bt_l, btu_l
        self.bt_g, self.btu_g, self.n_g = bt_l * W, btu_l * W, n_l * W
        nk, nf = 2 * n_l * K, 2 * n_l * FEAT_DIM
        self.pack_len = nf + bt_l
        alias = isinstance(self.comm, NoOpComm)
        self.pack = self._pack[:self.pack_len]
        self.recv = self.pack if alias else self._recv[:W * self.pack_len]
        self.logits_l = self._logits_loc[:nk].view(2, n_l, K)
        self.feat_l = self.pack[:nf].view(2, n_l, FEAT_DIM)
        self.labels_f = self.pack[nf:]
        self.logits_g = self._logits_g[:2 * self.n_g * K].view(2, self.n_g, K)
        self.feat_g = self._feat_g[:2 * self.n_g * FEAT_DIM].view(2, self.n_g, FEAT_DIM)
        self.labels_g = self._labels_g[:self.bt_g]
        self.probs_l = self._probs_l[:4 * btu_l * K].view(4, btu_l, K)
        self.probs_g = self.probs_l.view(1, 4, btu_l, K) if alias else self._probs_g[:W * 4 * btu_l * K].view(W, 4, btu_l, K)
        self.dlogits_l = self._dlogits_l[:nk].view(2, n_l, K)
        self.dfeat_l = self._dfeat_l[:nf].view(2, n_l, FEAT_DIM)
        # world 1: the column-side gradient partial IS this rank's slice of dfeat
        self.dfw_part = self.dfeat_l[1, bt_l:] if alias else self._dfw_part[:self.btu_g * FEAT_DIM].view(self.btu_g, FEAT_DIM)
        self.cshard = _lib.Shard(self.bt_g, self.btu_g, self.rank * bt_l, bt_l, self.rank * btu_l, btu_l)
        self._bound = (bt_l, btu_l)
        self._args = None           # the stage calls' ready-made argument tuples (see _stage_args)
        self._io = None             # cmlpl_dist_step's records (see _step_io)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _banks(self) -> _lib.Banks:
        # (one record, kept: only the two pointers change from step to step -- rebuilding the structs of a stage call
        #  on every step was a third of the sharded step's host time)
        b = self.__dict__.get("_c_banks")
        if b is None:
            b = self._c_banks = _lib.Banks()
            for i in range(2):
                b.d_feats[i] = self.bank_feats[i].data_ptr()
                b.d_probs[i] = self.bank_probs[i].data_ptr()
            b.Q = self.Q
        b.ptr[0], b.ptr[1] = self.ptr
        return b

    def _stage_args(self):
        """The seven stage calls' argument tuples for the bound shard, built ONCE: every pointer into the engine's own
        buffers as a ready-made ctypes instance (ctypes passes instances through without conversion), what changes from
        step to step in small ctypes cells that the stages update in place.  Marshalling fourteen arguments per call,
        seven calls per step, was 90 of the sharded step's ~175 us of host time -- more than the device leaves room for
        at a rank's shard (profiles/r06_final_dist_overhead.txt)."""
        vp, P = C.c_void_p, lambda t: C.c_void_p(t.data_ptr())
        c = self._cells = dict(step=C.c_uint64(0), smooth=C.c_int(0), adap=C.c_float(0.0), dm=vp(None), st=vp(None),
                               adam_t=C.c_int64(1), scal=vp(None), batch=_lib.Batch())
        sh, hp, bt, sd = C.byref(self.cshape), C.byref(self._chp), C.byref(c["batch"]), C.byref(self.cshard)
        banks, g = C.byref(self._banks()), C.byref(self._gathered())
        ws, wsn = P(self.workspace), C.c_size_t(self.workspace.numel())
        lws, lwsn = P(self.loss_ws), C.c_size_t(self.loss_ws.numel())
        one, seed, live = C.c_int(1), C.c_uint64(self.seed), C.c_int64(self.live)
        par, pk = P(self.params), P(self.packed)
        lib = self.lib
        self._args = dict(
            spectral=(lib.cmlpl_forward_spectral, (sh, hp, bt, sd, par, seed, c["step"], P(self.feat_l), P(self.labels_f), ws, wsn, c["st"])),
            spatial=(lib.cmlpl_forward_spatial, (sh, hp, bt, sd, par, pk, c["dm"], one, seed, c["step"], P(self.logits_l), ws, wsn, c["st"])),
            phase1=(lib.cmlpl_loss_phase1_g, (sh, sd, g, banks, c["smooth"], c["adap"], hp, P(self.dlogits_l), P(self.dfeat_l),
                                              P(self.probs_l), lws, lwsn, c["st"])),
            phase2=(lib.cmlpl_loss_phase2_g, (sh, sd, g, banks, c["smooth"], c["adap"], hp, P(self.probs_g), C.c_int(self.btu_l),
                                              c["scal"], P(self.dfeat_l), P(self.dfw_part), lws, lwsn, c["st"])),
            backward_data=(lib.cmlpl_backward_data, (sh, hp, bt, sd, par, pk, c["dm"], one, seed, c["step"], P(self.dlogits_l),
                                                     ws, wsn, c["st"])),
            backward_weights=(lib.cmlpl_backward_weights, (sh, hp, bt, sd, par, pk, c["dm"], one, seed, c["step"], P(self.dlogits_l),
                                                           P(self.dfeat_l), P(self.grads), live, ws, wsn, c["st"])),
            update=(lib.cmlpl_adam_step, (sh, C.c_int(2), par, C.c_int64(self.P), P(self.grads), live, P(self.m), P(self.v),
                                          c["adam_t"], hp, pk, c["st"])))
        return self._args

    def _call(self, name):
        fn, args = (self._args or self._stage_args())[name]
        self._cells["st"].value = torch.cuda.current_stream(self.device).cuda_stream
        rc = fn(*args)
        if rc != 0:
            raise _lib.CmlplError(fn.__name__, rc)

    # ------------------------------------------------------------------ stages (no communication inside)
    def _begin_step(self, XPl, Xl, Y, XPu, Xu, epoch, batch_index, noise=None, dropmask=None, apply_update=True,
                    lab_idx=None, unl_idx=None):
        """checks, the shard's views, and what changes from step to step written into the kept ctypes cells"""
        s = self.shape
        g = self._graph() if self._graph is not None else None
        if g is not None and g.pending > 0:
            raise RuntimeError(f"{g.pending} programmed graph replays are pending: launch them (or program() anew) "
                               "before an eager step")
        # (lab_idx / unl_idx: THIS rank's rows as indices into the resident splits, see TrainEngine.step)
        bt_l, btu_l = self._check_rows(XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx)
        if btu_l > self.btu_max:
            raise ValueError(f"shard {bt_l}+{btu_l} outside the engine's capacity {self.bt_max}+{self.btu_max} per rank")
        if self.Q < (bt_l + btu_l) * self.world:
            raise ValueError("bank smaller than the global batch")
        self._bind(bt_l, btu_l)
        n_l = bt_l + btu_l
        self._ensure_packed(self._stream())
        keep = None
        noise8 = None
        if noise is not None:
            keep = (C.c_void_p * 8)(*[t.data_ptr() for t in noise])
            noise8 = C.cast(keep, C.POINTER(C.c_void_p))
        if dropmask is not None:
            _chk_f32(dropmask, (2, n_l, s.cls_in), "dropmask")
        self._cur_row = self.step_count % self.hist_rows
        if self._args is None:
            self._stage_args()
        c = self._cells
        b = c["batch"]          # (updated in place: the stage calls hold a reference to this record)
        b.d_xpl, b.d_xl, b.d_xpu, b.d_xu, b.d_labels = XPl.data_ptr(), Xl.data_ptr(), XPu.data_ptr(), Xu.data_ptr(), Y.data_ptr()
        b.noise8, b.bt, b.btu = noise8, bt_l, btu_l
        b.d_lab_idx = None if lab_idx is None else lab_idx.data_ptr()
        b.d_unl_idx = None if unl_idx is None else unl_idx.data_ptr()
        c["step"].value = self.step_count
        c["smooth"].value = 1 if self.hp.smooth_gate(epoch, batch_index) else 0
        c["adap"].value = float(self.hp.thr * self.hp.adap_thr(epoch))
        c["dm"].value = None if dropmask is None else dropmask.data_ptr()
        c["adam_t"].value = self.adam_t + 1
        c["scal"].value = self.scalar_hist.data_ptr() + 64 * self._cur_row       # this step's row of the logging ring
        self._banks()           # (the record's pointers: ptr[] of this step)
        self._ctx = dict(apply_update=apply_update, keep=(keep, XPl, Xl, Y, XPu, Xu, noise, lab_idx, unl_idx, dropmask))

    def stage_spectral(self, *args, **kw):
        self._begin_step(*args, **kw)
        # the spectral branch of both networks: augmented spectra -> feat_spe -> ReLU -> L2 norm.  feat | labels land
        # directly in this rank's block of the exchange buffer; their all-gather starts behind this stage
        self._call("spectral")

    def stage_spatial(self):
        # augmentation of the patches + the convolution stack + head of both networks -> this rank's logits; the raw rows
        # are handed over as they are (the fused forward leaves the augmented rows in the workspace for the backward
        # stages: no other forward may use this workspace in between)
        self._call("spatial")

    def _gathered(self) -> _lib.Gathered:
        g = self.__dict__.get("_c_gathered")
        if g is None or self._c_gathered_for != self._bound:
            g = self._c_gathered = _lib.Gathered(self.recv.data_ptr(), self.logits_l.data_ptr(), self.world, self.bt_l, self.btu_l)
            self._c_gathered_for = self._bound
        return g

    def stage_phase1(self):
        # the loss kernels read the global rows where the all-gather left them (rank-major blocks): no re-ordering copy
        self._unpacked = False
        self._call("phase1")

    def stage_phase2(self):
        self._call("phase2")

    def stage_backward_data(self):
        # everything of the backward that needs dlogits alone (the reduce-scatter of the column-side feature gradient is
        # still in flight): data-gradient chain, conv0 and 3x3 weight-gradient partials
        self._call("backward_data")

    def stage_backward_weights(self):
        # dfeat is complete (this rank's slice of the reduce-scatter has arrived): dy takes its share, partials are
        # reduced, classifier / feat_spe weight gradients -> the gradient bucket
        self._call("backward_weights")

    def stage_update(self):
        if self._ctx["apply_update"]:
            self._call("update")
        self._end_step()

    def _end_step(self):
        if self._ctx["apply_update"]:
            self.adam_t += 1
        p0 = (self.ptr[0] + self.hp.bank_step) % self.Q                 # train.py:234,237
        self.ptr = [p0, (p0 + self.hp.bank_step) % self.Q]
        self.step_count += 1
        self._last_n = self.bt_l + self.btu_l
        self._ctx = None

    # the collectives that follow a stage: (kind, output, input, key).  key None: synchronous (on the critical path);
    # otherwise issued asynchronously and waited for in front of the stage that names the key in waits_before
    def exchange_after(self, stage: str):
        if stage == "spectral":         # the embeddings (+ labels) of every rank: runs under the convolution forward
            return [("all_gather", self.recv, self.pack, "feat")]
        if stage == "phase1":
            return [("all_gather", self.probs_g, self.probs_l, None)]
        if stage == "phase2":           # backward of "gather the keys": sum the partials, keep this rank's rows;
            return [("reduce_scatter", self.dfeat_l[1, self.bt_l:], self.dfw_part, "dfw")]   # runs under the convolution backward
        if stage == "backward_weights":  # one flat bucket: the live tensors of both networks
            return [("all_reduce", self.grads, None, None)]
        return []

    def waits_before(self, stage: str):
        if stage == "phase1":           # similarity tiles against every rank's keys, labels for the bank write
            return ["feat"]
        if stage == "backward_weights":  # dy's feature-gradient share, then feat_spe's weight gradient
            return ["dfw"]
        return []

    # ------------------------------------------------------------------ the step
    def step(self, XPl, Xl, Y, XPu, Xu, epoch: int, batch_index: int, noise: Optional[Sequence[torch.Tensor]] = None,
             dropmask: Optional[torch.Tensor] = None, apply_update: bool = True, lab_idx=None, unl_idx=None) -> None:
        """Per-rank inputs: this rank's bt/W labelled and btu/W unlabelled rows (noise / dropmask, when given,
        are this rank's slices too) -- or, with lab_idx / unl_idx, the resident splits and this rank's row indices."""
        coll = self._native_comm()
        if coll is False:           # a communicator only Python can drive (torch.distributed, the tests' stand-ins)
            drive_step(self, self.comm, XPl, Xl, Y, XPu, Xu, epoch, batch_index, noise, dropmask, apply_update,
                       lab_idx=lab_idx, unl_idx=unl_idx)
            return
        # the whole step as ONE C call: stages and collectives enqueued back to back (cmlpl_dist_step, csrc/dist.hip)
        self._begin_step(XPl, Xl, Y, XPu, Xu, epoch, batch_index, noise, dropmask, apply_update, lab_idx=lab_idx, unl_idx=unl_idx)
        io, a, c = self._step_io()
        io.batch, io.banks = c["batch"], self._c_banks
        a.step, a.adam_t, a.d_dropmask = c["step"].value, c["adam_t"].value, c["dm"].value
        a.smooth, a.adap_mask, a.apply_update, a.scalars_row = c["smooth"].value, c["adap"].value, 1 if apply_update else 0, self._cur_row
        self._unpacked = False
        _lib.check("cmlpl_dist_step", self.lib.cmlpl_dist_step(
            C.byref(self.cshape), C.byref(self._chp), C.byref(io), C.byref(a), coll, self._stream()))
        self._end_step()

    def _native_comm(self):
        """the communicator as cmlpl_dist_step takes it: a cmlpl_collectives record (RcclComm), None for the aliased
        one-rank step -- or False: not available (or switched off: ``native_step = False`` / CMLPL_DIST_NATIVE=0)"""
        if not self.native_step:
            return False
        if isinstance(self.comm, NoOpComm):
            return None
        nat = getattr(self.comm, "native", None)
        return C.byref(nat(self.async_exchanges)) if nat is not None else False

    def _step_io(self):
        """cmlpl_dist_io of the bound shard in by-value mode + the per-step record, built once per shard"""
        if self._args is None:
            self._stage_args()
        if self._io is None:
            io = _lib.DistIO()
            io.shard, io.gathered = self.cshard, self._gathered()
            io.d_params, io.d_m, io.d_v, io.d_packed = self.params.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.packed.data_ptr()
            io.d_grads, io.grad_stride = self.grads.data_ptr(), self.live
            io.d_logits_l, io.d_feat_l, io.d_labels_f = self.logits_l.data_ptr(), self.feat_l.data_ptr(), self.labels_f.data_ptr()
            io.d_dlogits, io.d_dfeat = self.dlogits_l.data_ptr(), self.dfeat_l.data_ptr()
            io.d_probs_l, io.d_probs_g, io.probs_shard_rows = self.probs_l.data_ptr(), self.probs_g.data_ptr(), self.btu_l
            io.d_scalars, io.d_dfeat_w_partial = self.scalar_hist.data_ptr(), self.dfw_part.data_ptr()
            io.d_workspace, io.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel()
            io.d_loss_workspace, io.loss_workspace_bytes = self.loss_ws.data_ptr(), self.loss_ws.numel()
            io.seed = self.seed
            self._io = (io, _lib.DistStepArgs())
        return self._io[0], self._io[1], self._cells

    def capture(self, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt: int, btu: int, capacity: int = 1024) -> "DistStepGraph":
        """The sharded step as FIVE captured graphs, one per stage, with the four collectives eager between them
        (cmlpl_dist_stage_graph_create); ``bt`` / ``btu`` are THIS rank's rows, ``lab_idx`` / ``unl_idx`` the resident
        index buffers the step offsets point into.  See DistStepGraph."""
        return DistStepGraph(self, XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt, btu, capacity)

    def outputs(self, gathered_logits=None):
        """(logits, feat) of the GLOBAL batch of the last step, [2][n_g][..], in global row order [labelled of all
        ranks ; unlabelled of all ranks] (re-ordered on demand from the gathered blocks; the step itself never
        makes this copy and never gathers the logits: they are gathered HERE, a collective every rank must join --
        or handed in as ``gathered_logits``, the ranks' logits_l back to back)."""
        if not getattr(self, "_unpacked", False):
            nk = self.logits_l.numel()
            if gathered_logits is not None:          # (a harness that moved the blocks itself: tests' lockstep ranks)
                recv_z = gathered_logits.reshape(-1)
                assert recv_z.numel() == self.world * nk
            elif isinstance(self.comm, NoOpComm):
                recv_z = self.logits_l
            else:                                    # the step never gathers logits: do it for this call
                recv_z = self._recv_z[:self.world * nk]
                self.comm.all_gather(recv_z, self.logits_l)
            _lib.check("cmlpl_dist_unpack", self.lib.cmlpl_dist_unpack(
                C.byref(self.cshape), self.world, self.bt_l, self.btu_l, self.recv.data_ptr(), recv_z.data_ptr(),
                self._logits_g.data_ptr(), self._feat_g.data_ptr(), self._labels_g.data_ptr(), self._stream()))
            self._unpacked = True
        return self.logits_g, self.feat_g

    def read_scalars(self):
        t = self.scalars.clone()
        self.comm.all_reduce(t)             # shares are additive (finalize_kernel)
        return dict(zip(SCALAR_NAMES, t.tolist()))

    def loss_row(self):
        t = self.scalars.clone()
        self.comm.all_reduce(t)
        return t[:5].tolist()

    def _reduce_rows(self, rows):
        rows = rows.contiguous()
        self.comm.all_reduce(rows)          # one collective per printed window
        return rows


class DistStepGraph:
    """The sharded training step replayed from seven hipGraphs (spectral | spatial | phase 1 | phase 2 | backward data |
    backward weights | update) with the step's four collectives issued eagerly between them, exactly where and how
    ``drive_step`` issues them: the embedding all-gather asynchronously behind the spectral stage (waited for in front of
    phase 1), the probability all-gather, the reduce-scatter of the column-side feature gradient asynchronously behind
    phase 2 (waited for in front of the backward-weights stage), the all-reduce of the gradient bucket.  Everything that
    changes from step to step comes from the same device table as the single-GPU ``StepGraph`` (``program()``: (epoch,
    batch_index, lab_off, unl_off) with THIS rank's offsets into the index buffers); a replay costs the host seven graph
    launches and four torch.distributed calls instead of seven marshalled stage calls (scripts/dist_overhead.py).
    Bit-identical to the eager sharded step."""

    def __init__(self, eng: "DistTrainEngine", XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx, bt: int, btu: int, capacity: int = 1024):
        import numpy as np
        self.eng, self.bt, self.btu, self.capacity = eng, int(bt), int(btu), int(capacity)
        if lab_idx is None or unl_idx is None:
            raise ValueError("a captured step reads its rows through index buffers")
        for name, t in (("lab_idx", lab_idx), ("unl_idx", unl_idx)):
            if t.dtype != torch.int64 or t.dim() != 1 or not t.is_cuda or not t.is_contiguous():
                raise ValueError(f"{name}: need a contiguous int64 cuda vector")
        if lab_idx.shape[0] < bt or unl_idx.shape[0] < btu:
            raise ValueError("index buffers shorter than one shard")
        if eng.step_count == 0:
            raise RuntimeError("run one eager DistTrainEngine.step() before capturing (kernel attributes are set lazily)")
        eng._check_rows(XPl, Xl, Y, XPu, Xu, lab_idx[:bt], unl_idx[:btu])
        TrainEngine.check_index_range(lab_idx, XPl.shape[0], "lab_idx")
        TrainEngine.check_index_range(unl_idx, XPu.shape[0], "unl_idx")
        eng._bind(self.bt, self.btu)
        self._keep = (XPl, Xl, Y, XPu, Xu, lab_idx, unl_idx)
        self.n_lab_idx, self.n_unl_idx = int(lab_idx.shape[0]), int(unl_idx.shape[0])
        dev = eng.device
        self.table = torch.zeros((self.capacity + 2) * 64, dtype=torch.uint8, device=dev)
        self.cursor = torch.ones(1, dtype=torch.int32, device=dev)
        self.host = torch.zeros((self.capacity + 2) * 64, dtype=torch.uint8).pin_memory()
        self.rows = self.host.numpy().view(np.dtype(_lib.DYN_DTYPE))[1:]
        self.pending = 0
        st = eng._stream()
        eng._ensure_packed(st)
        io = _lib.DistIO()
        io.batch = _lib.Batch(XPl.data_ptr(), Xl.data_ptr(), XPu.data_ptr(), Xu.data_ptr(), Y.data_ptr(), None, self.bt, self.btu,
                              lab_idx.data_ptr(), unl_idx.data_ptr())
        io.shard = eng.cshard
        io.gathered = eng._gathered()
        io.banks = eng._banks()
        io.d_params, io.d_m, io.d_v, io.d_packed = eng.params.data_ptr(), eng.m.data_ptr(), eng.v.data_ptr(), eng.packed.data_ptr()
        io.d_grads, io.grad_stride = eng.grads.data_ptr(), eng.live
        io.d_logits_l, io.d_feat_l, io.d_labels_f = eng.logits_l.data_ptr(), eng.feat_l.data_ptr(), eng.labels_f.data_ptr()
        io.d_dlogits, io.d_dfeat = eng.dlogits_l.data_ptr(), eng.dfeat_l.data_ptr()
        io.d_probs_l, io.d_probs_g, io.probs_shard_rows = eng.probs_l.data_ptr(), eng.probs_g.data_ptr(), self.btu
        io.d_scalars = eng.scalar_hist.data_ptr()
        io.d_dfeat_w_partial = eng.dfw_part.data_ptr()
        io.d_workspace, io.workspace_bytes = eng.workspace.data_ptr(), eng.workspace.numel()
        io.d_loss_workspace, io.loss_workspace_bytes = eng.loss_ws.data_ptr(), eng.loss_ws.numel()
        io.seed = eng.seed
        io.d_dyn_table, io.d_dyn_cursor = self.table.data_ptr(), self.cursor.data_ptr()
        self._io = io
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        self.handles = {}
        with torch.cuda.stream(cap):
            for name in eng.STAGES:
                h = C.c_void_p()
                _lib.check("cmlpl_dist_stage_graph_create", eng.lib.cmlpl_dist_stage_graph_create(
                    C.byref(eng.cshape), C.byref(eng._chp), C.byref(io), _lib.STAGE_IDS[name], C.c_void_p(cap.cuda_stream),
                    C.byref(h)))
                self.handles[name] = h
        torch.cuda.current_stream(dev).wait_stream(cap)

    def validate_indices(self) -> None:
        """range check of the two index buffers (synchronising): after the caller re-filled them in place"""
        XPl, _, _, XPu, _, lab_idx, unl_idx = self._keep
        TrainEngine.check_index_range(lab_idx, XPl.shape[0], "lab_idx")
        TrainEngine.check_index_range(unl_idx, XPu.shape[0], "unl_idx")

    def program(self, steps) -> None:
        """as StepGraph.program: (epoch, batch_index, lab_off, unl_off) of the next replays, offsets = this rank's"""
        from .engine import StepGraph
        StepGraph.program(self, steps)

    def launch_stage(self, name: str) -> None:
        """enqueue ONE stage's graph (the lockstep test harness performs the exchanges itself; ``launch`` is the product path)"""
        eng = self.eng
        if self.pending < 1:
            raise RuntimeError("no programmed step left: call program() first")
        if name == eng.STAGES[0]:
            eng._bind(self.bt, self.btu)      # (an eager short batch in between re-bound the views; same pointers for the same shard)
            eng._ensure_packed(eng._stream())
            eng._cur_row = eng.step_count % eng.hist_rows
            eng._unpacked = False
        _lib.check("cmlpl_step_graph_launch", eng.lib.cmlpl_step_graph_launch(self.handles[name], eng._stream()))
        if name == "update":                  # host copy of the step bookkeeping (train.py:234,237)
            eng.adam_t += 1
            p0 = (eng.ptr[0] + eng.hp.bank_step) % eng.Q
            eng.ptr = [p0, (p0 + eng.hp.bank_step) % eng.Q]
            eng.step_count += 1
            eng._last_n = self.bt + self.btu
            self.pending -= 1

    def launch(self) -> None:
        """one replay = one sharded training step (asynchronous): seven graph launches, four collectives -- two of them
        in flight under the convolution stages, as in drive_step"""
        eng = self.eng
        pending = {}
        for name in eng.STAGES:
            for key in eng.waits_before(name):
                h = pending.pop(key)
                if h is not None:
                    h.wait()
            self.launch_stage(name)
            for spec in eng.exchange_after(name):
                h = issue_exchange(eng.comm, spec)
                if spec[3] is not None:
                    pending[spec[3]] = h
        assert not pending

    def close(self) -> None:
        for name, h in list(self.handles.items()):
            if h:
                self.eng.lib.cmlpl_step_graph_destroy(h)
        self.handles = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
