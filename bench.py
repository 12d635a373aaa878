#!/usr/bin/env python3
"""bench.py -- CMLPL training-step throughput on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload B2|B3|P|B4|B5] [--global-batch BT+BTU]

N > 1 works both ways: typed as above, the parent starts N ranks itself (cmlpl_amd/launch.py: fresh child
processes with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, before anything touches the GPU) and relays rank 0's
line; launched under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks are
used as they come.  Default: 128+128 rows PER GPU ("scaling": "weak").  `--global-batch 512+512` fixes the
GLOBAL batch and shards it over the ranks ("scaling": "strong"): BASELINE configs[2] is `--workload B3`
(= B2 shape, global 512+512 -> 64+64 per rank at 8 GPUs), configs[4] is `--workload B5 --global-batch 64+512`
(-> 8+64 per rank).

A "step" is one pass of the hot path (reference train.py:150-278: noise augmentation, two BaseNet2
forwards, loss block, bank write, two backwards, two Adam steps) over one synthetic batch that is
already resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {   # name: (C, H, W, bands, K)   -- SURVEY.md section 8d
    "B2": (103, 11, 11, 103, 9),    # BASELINE.json configs[1]: PaviaU 11x11x103, batch 256
    "B3": (103, 11, 11, 103, 9),    # BASELINE.json configs[2]: same shape, GLOBAL batch 512+512 sharded over the GPUs
    "P": (60, 20, 20, 103, 9),      # the reference's own (PCA-60, 20x20) shape
    "B4": (200, 11, 11, 200, 16),   # Indian-Pines-shaped
    "B5": (48, 15, 15, 48, 20),     # Houston2018-shaped
}
FP32_MFMA_PEAK_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0      # same guide, "BF16/F16 ~2.5 PF dense"
# The 3x3 convolutions compute fp32 products as six bf16 MFMAs (three exact bf16 pieces per operand, fp32 accumulate;
# cmlpl_amd/csrc/conv3x3.hip): their ceiling in fp32-equivalent FLOP/s is the bf16 peak / 6.
SPLIT_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
SPLIT2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0      # fp32 as two fp16 pieces: three 16-bit MFMAs per product (f16 rate = bf16 rate)


def blended_peak(segments):
    """fp32-equivalent MFMA peak of a kernel that mixes f32-input and split-bf16 MFMA segments:
    total FLOPs / sum(FLOPs_i / peak_i).  segments: {"f32": flops, "split": flops (three bf16 pieces, six MFMAs per
    product), "split2": flops (two fp16 pieces, three MFMAs per product)}."""
    tot = segments.get("f32", 0.0) + segments.get("split", 0.0) + segments.get("split2", 0.0)
    floor_s = (segments.get("f32", 0.0) / (FP32_MFMA_PEAK_TFLOPS * 1e12) + segments.get("split", 0.0) / (SPLIT_PEAK_TFLOPS * 1e12) +
               segments.get("split2", 0.0) / (SPLIT2_PEAK_TFLOPS * 1e12))
    return (tot / floor_s / 1e12) if floor_s > 0 else FP32_MFMA_PEAK_TFLOPS
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")    # written by scripts/pmc_summary.py --json
CONV1_KERNELS = {     # labels of the unfused launches; the fused ones are named where the calibration window finds them
    "conv1_fwd": "conv3x3_kernel<0> (conv1 forward: 3x3 conv + bias + residual + ReLU + avgpool, both networks)",
    "conv1_dgrad": "conv3x3_kernel<1> (conv1 data gradient, both networks)",
    "conv1_wgrad": "wgrad3b_kernel (conv1 weight gradient, split-bf16, both networks)",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="B2", choices=sorted(WORKLOADS))
    ap.add_argument("--bt", type=int, default=128, help="labelled rows per GPU (weak scaling)")
    ap.add_argument("--btu", type=int, default=128, help="unlabelled rows per GPU (weak scaling)")
    ap.add_argument("--global-batch", default=None, metavar="BT+BTU",
                    help="fix the GLOBAL batch (labelled+unlabelled) and shard it over the GPUs: strong scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true",
                    help="N > 1: every rank shards ONE common global batch; rank 0 then runs the same global batch "
                         "through the single-GPU engine and the first step's logged scalars must agree (1e-4 relative) "
                         "-- exits non-zero otherwise")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel time table to stderr")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.workload == "B3" and args.global_batch is None:
        args.global_batch = "512+512"
    return args


def per_rank_batch(args, world):
    """(bt, btu) rows per rank and the scaling mode.  A global batch must divide evenly: equal shards are what
    makes local means / W equal the global mean."""
    if args.global_batch is None:
        return args.bt, args.btu, "weak"
    try:
        gbt, gbtu = (int(v) for v in args.global_batch.split("+"))
    except ValueError:
        raise SystemExit("--global-batch wants BT+BTU, e.g. 512+512")
    if gbt < world or gbtu < world or gbt % world or gbtu % world:
        raise SystemExit(f"--global-batch {args.global_batch} does not shard evenly over {world} GPUs")
    return gbt // world, gbtu // world, "strong"


def synth(shape, bt, btu, seed, device):
    """XP, X ~ N(0,1), Y ~ U{0..K-1} from torch.Generator(seed) (reference seed 1088, train.py:50)."""
    import torch
    C, H, W, bands, K = shape
    g = torch.Generator().manual_seed(seed)
    d = dict(XPl=torch.randn(bt, C, H, W, generator=g), Xl=torch.randn(bt, bands, generator=g),
             Y=torch.randint(0, K, (bt,), generator=g),
             XPu=torch.randn(btu, C, H, W, generator=g), Xu=torch.randn(btu, bands, generator=g))
    return {k: v.to(device) for k, v in d.items()}


def conv1_flops(shape, n, nets=2):
    """Algorithmic FLOPs of ONE 3x3 64->64 convolution pass (fwd, dgrad or wgrad) over n patches per net,
    dense-conv count as in SURVEY.md 8d: 2 * H*W * 64 * 576 per patch."""
    C, H, W, bands, K = shape
    return 2.0 * nets * n * H * W * 64 * 576


def recorded_traffic(workload, n_local):
    """HBM bytes per launch of the conv1 kernels as measured by rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE, separate
    passes; scripts/pmc_summary.py --json writes the file).  The record carries the hash of the kernel sources it
    was measured on: if the sources changed since, or the workload differs, the figure is stale -> None."""
    try:
        rec = json.load(open(TRAFFIC_FILE))
        from cmlpl_amd.build_ext import source_hash
        if rec.get("source_hash") != source_hash() or rec.get("workload") != workload or rec.get("n_local") != n_local:
            return {}, rec.get("profile")
        return {k: float(v) for k, v in rec["bytes_per_launch"].items()}, rec.get("profile")
    except Exception:
        return {}, None


def _host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:  # container CPU share (cgroup v2 quota), if any: oversubscribing it makes the baseline meaningless
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return min(cores, int(os.environ.get("CMLPL_CPU_THREADS", "32")))   # torch intra-op scaling flattens beyond this


def cpu_baseline(shape, bt, btu, budget_s=12.0, budget_1t_s=7.0, budget_b1_s=5.0):
    """The oracle (CPU restatement of the reference step, verified against the reference's own outputs)
    timed on this host's cores on the same workload; bounded sample.  All host cores, then one thread
    (SURVEY.md 8d asks for both figures)."""
    import torch
    from oracle import cmlpl_oracle as O
    s = O.NetShape(*shape)
    hp = O.HyperParams()
    st = O.StepState.create(s, O.closed_form_params(s, 1), O.closed_form_params(s, 2), bt, hp)
    batches = [O.synthetic_batch(s, bt, btu, 1088 + i) for i in range(2)]

    def sample(threads, budget, warm):
        torch.set_num_threads(threads)
        times = []
        t_start = time.perf_counter()
        i = 0
        while True:
            b = batches[i % 2]
            t0 = time.perf_counter()
            # the reference draws noise and the dropout mask inside the step (train.py:157-182, models.py:148)
            noise = [torch.randn_like(t) for t in b["noise"]]
            keep = 1.0 - hp.dropout
            dm = [(torch.rand(bt + btu, s.cls_in) < keep).float() / keep for _ in range(2)]
            O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], noise, dm, 1, i, hp)
            times.append(time.perf_counter() - t0)
            i += 1
            if i >= warm + 1 and (time.perf_counter() - t_start > budget or i >= 200):
                break
        times = times[warm:]
        return sorted(times)[len(times) // 2], len(times)

    cores = _host_cores()
    med, k = sample(cores, budget_s, 2)
    med1, k1 = sample(1, budget_1t_s, 1)
    out = {"value": (bt + btu) / med, "unit": "patches/s", "cores": cores, "kind": "port",
           "sample": f"{k} steps of the same {bt}+{btu} workload after 2 warm-up steps, median "
                     f"{med * 1e3:.1f} ms/step, PyTorch-CPU {torch.__version__}, {cores} threads",
           "value_1thread": (bt + btu) / med1,
           "sample_1thread": f"{k1} steps after 1 warm-up step, median {med1 * 1e3:.1f} ms/step, 1 thread"}
    # BASELINE.json configs[0] / BASELINE.md B1: the reference's own CPU-runnable case (its 60x20x20 shape, 32 + 32 rows)
    sB = O.NetShape(60, 20, 20, 103, 9)
    stB = O.StepState.create(sB, O.closed_form_params(sB, 1), O.closed_form_params(sB, 2), 32, hp)
    bB = [O.synthetic_batch(sB, 32, 32, 2088 + i) for i in range(2)]
    torch.set_num_threads(cores)
    tB = []
    t_start = time.perf_counter()
    i = 0
    while True:
        b = bB[i % 2]
        t0 = time.perf_counter()
        noise = [torch.randn_like(t) for t in b["noise"]]
        dm = [(torch.rand(64, sB.cls_in) < 1.0 - hp.dropout).float() / (1.0 - hp.dropout) for _ in range(2)]
        O.train_step(stB, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], noise, dm, 1, i, hp)
        tB.append(time.perf_counter() - t0)
        i += 1
        if i >= 3 and (time.perf_counter() - t_start > budget_b1_s or i >= 200):
            break
    medB = sorted(tB[2:])[len(tB[2:]) // 2]
    out["b1"] = {"value": 64 / medB, "unit": "patches/s", "cores": cores,
                 "sample": f"BASELINE configs[0] (reference shape 60x20x20, 32+32 rows): {len(tB) - 2} steps after 2 "
                           f"warm-up steps, median {medB * 1e3:.1f} ms/step, {cores} threads"}
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` typed directly: start N ranks of this same program (no GPU call was made in
    this process), relay rank 0's JSON line, fail if any rank fails."""
    from cmlpl_amd.launch import spawn_ranks
    extra = {"CMLPL_FORCE_DIST": "1"} if args.gpus == 1 else None
    rc, out = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + list(argv), extra_env=extra)
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 1
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return rc


def check_against_single_rank(eng, shape, bt, btu, world, rank, device, hp):
    """--check: ONE global batch (common seed), sharded by sample as the data-parallel engine expects (rank r takes
    labelled rows [r bt, (r+1) bt) and unlabelled rows [r btu, (r+1) btu)); step 0 on all ranks, then the same global
    batch through the single-GPU TrainEngine on rank 0 from the same parameters and seed (the in-kernel noise / dropout
    streams are keyed by the GLOBAL sample index, so the two runs draw the same numbers).  The logged row and the
    Base1 losses must agree to 1e-4 relative; a mismatch ends the job with a non-zero code."""
    import torch
    from cmlpl_amd import NetShape, TrainEngine
    g = synth(shape, bt * world, btu * world, 4242, device)
    sl = lambda t, k: t[rank * k:(rank + 1) * k].contiguous()
    snap = [eng.params.clone(), eng.m.clone(), eng.v.clone(), eng.bank_feats.clone(), eng.bank_probs.clone()]
    state = (list(eng.ptr), eng.adam_t, eng.step_count)
    eng.step(sl(g["XPl"], bt), sl(g["Xl"], bt), sl(g["Y"], bt), sl(g["XPu"], btu), sl(g["Xu"], btu), epoch=1, batch_index=0)
    got = eng.read_scalars()                      # all-reduced: every rank calls it
    ok = True
    if rank == 0:
        ref = TrainEngine(NetShape(*shape), bt * world, btu * world, hp, device=device, seed=1088)
        ref.params.copy_(snap[0])
        ref._packed_dirty = True
        ref.step(g["XPl"], g["Xl"], g["Y"], g["XPu"], g["Xu"], epoch=1, batch_index=0)
        want = ref.read_scalars()
        for k in ("ctr_s", "total_s", "cls_s", "con_s", "acc", "total_w", "cls_w", "con_w"):
            if not abs(got[k] - want[k]) <= 1e-4 * abs(want[k]) + 1e-6:
                ok = False
        print(f"bench.py --check: {world} ranks {['MISMATCH', 'ok'][ok]}: sharded {got} vs single-rank {want}", file=sys.stderr)
        del ref
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    eng.comm.all_reduce(flag)                     # rank 0's verdict reaches every rank (the others contribute 1)
    # restore the state the timed run starts from
    for dst, src in zip((eng.params, eng.m, eng.v, eng.bank_feats, eng.bank_probs), snap):
        dst.copy_(src)
    eng.ptr, eng.adam_t, eng.step_count = list(state[0]), state[1], state[2]
    eng._packed_dirty = True
    if float(flag.item()) < world - 0.5:
        raise SystemExit("bench.py --check: the sharded step does not reproduce the single-rank step")


def run_rank(args):
    # RCCL / HIP runtime banners go to stdout; the contract is ONE JSON line there.  Keep the real stdout
    # aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    bt, btu, scaling = per_rank_batch(args, world)
    # CMLPL_ONE_GPU=1 + CMLPL_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a one-GPU box (all ranks on cuda:0;
    # RCCL refuses two ranks on one device).  Never set by the driver; the numbers of such a run mean nothing.
    one_gpu = bool(os.environ.get("CMLPL_ONE_GPU"))
    backend = os.environ.get("CMLPL_DIST_BACKEND", "nccl")
    device = torch.device("cuda:0" if one_gpu else f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from cmlpl_amd import HyperParams, NetShape, TrainEngine, _lib
    shape = WORKLOADS[args.workload]
    hp = HyperParams()
    dist = None
    if world > 1 or os.environ.get("CMLPL_FORCE_DIST"):
        if not os.environ.get("MASTER_ADDR"):
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29513", RANK="0", WORLD_SIZE="1")
        from cmlpl_amd.distributed import DistStartupError, DistTrainEngine, init_distributed
        try:     # checked start-up: one device per local rank, rendezvous and first collectives under a watchdog
            dist = init_distributed(backend, device, timeout_s=float(os.environ.get("CMLPL_DIST_TIMEOUT", "300")),
                                    one_gpu=one_gpu)
        except DistStartupError as e:
            raise SystemExit(f"bench.py: {e}")
        eng = DistTrainEngine(NetShape(*shape), bt, btu, hp, device=device, seed=1088)
    else:
        eng = TrainEngine(NetShape(*shape), bt, btu, hp, device=device, seed=1088)
    eng.init_params_default(1088)
    if args.check and world > 1:
        check_against_single_rank(eng, shape, bt, btu, world, rank, device, hp)
    batches = [synth(shape, bt, btu, 1088 + 7919 * rank + i, device) for i in range(4)]
    lib = _lib.load()

    def run(k, first_index):
        for i in range(k):
            b = batches[(first_index + i) % len(batches)]
            eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], epoch=1, batch_index=first_index + i)

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    run(args.warmup, 0)
    barrier()
    # Which kernel is the dominant one is decided by measurement: a short calibration window (untimed) brackets
    # the three conv1 launches with hipEvent pairs on the launch stream; the longest one is then bracketed inside
    # the timed region (one event pair per step keeps the perturbation of the step time below 1 %).
    import ctypes as C
    nk = len(_lib.KERNEL_NAMES)
    ms = (C.c_double * nk)()
    cnt = (C.c_int64 * nk)()
    ids = {k: _lib.KERNEL_NAMES.index(k) for k in CONV1_KERNELS}
    id_conv0 = _lib.KERNEL_NAMES.index("conv0_fwd")
    id_conv0w = _lib.KERNEL_NAMES.index("conv0_wgrad")
    id_conv2f = _lib.KERNEL_NAMES.index("conv2_fwd")
    id_conv2d = _lib.KERNEL_NAMES.index("conv2_dgrad")
    id_conv2w = _lib.KERNEL_NAMES.index("conv2_wgrad")
    calib_steps = 10
    _lib.check("cmlpl_timing_begin", lib.cmlpl_timing_begin(
        sum(1 << i for i in ids.values()) | (1 << id_conv0) | (1 << id_conv0w) | (1 << id_conv2f) | (1 << id_conv2d) | (1 << id_conv2w),
        7 * calib_steps + 8))
    run(calib_steps, args.warmup)
    barrier()
    _lib.check("cmlpl_timing_end", lib.cmlpl_timing_end(ms, cnt))
    calib = {k: ms[i] / max(cnt[i], 1) for k, i in ids.items()}
    # no separate conv0 launch => the forward kernel is the fused conv0 + conv1 one and carries both FLOP counts
    fused_fwd = cnt[id_conv0] == 0
    fused_bwd = cnt[id_conv0w] == 0
    # no separate conv2 launches either => they run in the tails of the fused per-sample kernels
    tail_fwd = fused_fwd and cnt[id_conv2f] == 0
    head_bwd = fused_bwd and cnt[id_conv2d] == 0
    wgrad_pair = cnt[id_conv2w] == 0        # conv2's weight gradient rides in conv1's launch
    dom_name = max(calib, key=calib.get)
    dom_id = ids[dom_name]
    _lib.check("cmlpl_timing_begin", lib.cmlpl_timing_begin(1 << dom_id, args.steps + 8))
    thr0 = cgroup_throttle()
    t0 = time.perf_counter()
    run(args.steps, args.warmup + calib_steps)
    barrier()
    dt = time.perf_counter() - t0
    thr1 = cgroup_throttle()
    _lib.check("cmlpl_timing_end", lib.cmlpl_timing_end(ms, cnt))
    if dist is not None:
        t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    scal = eng.read_scalars()
    if not all(v == v and abs(v) != float("inf") for v in scal.values()) and not os.environ.get("CMLPL_BENCH_ALLOW_NONFINITE"):   # (ablation builds)
        raise SystemExit(f"bench.py: non-finite loss after the timed region: {scal}")

    n_local = bt + btu
    patches = n_local * world * args.steps
    dom_ms = ms[dom_id] / max(cnt[dom_id], 1)
    wgrad_split = os.environ.get("CMLPL_WGRAD3_B3", "1") != "0"
    c1 = conv1_flops(shape, n_local)
    # per kernel: algorithmic FLOPs by MFMA kind ("split" = three-piece bf16 operands, "f32" = f32-input MFMA)
    kseg = {"conv1_fwd": {"split": c1}, "conv1_dgrad": {"split": c1},
            "conv1_wgrad": {"split": c1} if wgrad_split else {"f32": c1}}
    labels = dict(CONV1_KERNELS)
    conv0_flops = 2.0 * 2 * n_local * shape[1] * shape[2] * shape[0] * 64                # 2*nets*n*HW*C*64
    if fused_fwd:
        kseg["conv1_fwd"]["split"] += conv0_flops                                       # conv0 stage: split-bf16 too
        kseg["conv1_fwd"]["f32"] = 0.0
        labels["conv1_fwd"] = ("conv3x3_kernel<FWD0> (conv0 1x1 + conv1 3x3 forward fused: conv + bias + residual + "
                               "ReLU + avgpool, both networks)")
    if fused_bwd:
        kseg["conv1_dgrad"]["split"] += conv0_flops                                     # conv0 weight gradient: split-bf16 too
        labels["conv1_dgrad"] = ("conv3x3_kernel<DGRAD0> (conv1 data gradient + conv0 weight gradient fused, "
                                 "both networks)")
    conv2_flops = 2.0 * 2 * n_local * (shape[1] // 2) * (shape[2] // 2) * 64 * 576    # dense 3x3 on the pooled map
    # which instantiation of the per-sample kernels this launch takes (conv3x3.hip): <MODE, 1, 1, waves, tiles per wave> --
    # eight waves x two tiles for windows of 129..256 pixels, eight waves x one tile when the grid fits the CUs, else four x two
    if shape[1] * shape[2] > 128:
        kvar = "8,2"
    elif 2 * n_local <= 256 and os.environ.get("CMLPL_KS8", "-1") != "0" or os.environ.get("CMLPL_KS8") == "1":
        kvar = "8,1"
    else:
        kvar = "4,2"
    # which products run on TWO fp16 pieces (three MFMAs per product) is the planners' decision: asked of the library
    # (conv1's tap loops, the weight gradients; on the general path also conv2's launches); the rest on three bf16 pieces
    cs_ = _lib.Shape(*shape)
    tp = lib.cmlpl_debug_two_piece(C.byref(cs_), 2, n_local)
    if tp < 0:
        raise SystemExit(f"bench.py: cmlpl_debug_two_piece: {tp}")
    if tp & 1:
        kseg["conv1_fwd"]["split"] -= c1; kseg["conv1_fwd"]["split2"] = c1
    if tp & 2:
        kseg["conv1_dgrad"]["split"] -= c1; kseg["conv1_dgrad"]["split2"] = c1
    if tail_fwd:
        kseg["conv1_fwd"]["split"] += conv2_flops                                       # tail: split-bf16 too (16x16x32 MFMA)
        labels["conv1_fwd"] = (f"conv3x3_kernel<2,1,1,{kvar}> (per-sample fused forward: augmentation + conv0 1x1 + conv1 3x3 + "
                               "ReLU/pool + conv2 3x3 + ReLU/pool + concat/dropout/classifier/L2-norm, both networks)")
    if head_bwd:
        kseg["conv1_dgrad"]["split"] += conv2_flops
        labels["conv1_dgrad"] = (f"conv3x3_kernel<3,1,1,{kvar}> (per-sample fused backward: head + conv2 data gradient + conv1 "
                                 "data gradient + conv0 weight gradient, both networks)")
    if wgrad_pair:
        key = "split" if wgrad_split else "f32"
        kseg["conv1_wgrad"][key] += conv2_flops
        # ... and the pair weight-gradient launch both its maps (planes of two fp16 pieces) when the step's 3x3 launches
        # leave the operands' maxima it scales by
        if wgrad_split and (tp & 4):
            kseg["conv1_wgrad"] = {"split2": kseg["conv1_wgrad"]["split"]}
        labels["conv1_wgrad"] = ("wgrad3b_pair_kernel (conv1 + conv2 weight gradients in one launch, both networks" +
                                 (", two-piece fp16 planes)" if wgrad_split and (tp & 4) else ")"))
    kflops = {k: sum(v.values()) for k, v in kseg.items()}
    kpeak = {k: blended_peak(v) for k, v in kseg.items()}
    traffic, traffic_src = recorded_traffic(args.workload, n_local)
    flops = kflops[dom_name]
    achieved = flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    out = {
        "metric": "HSI patches/sec per training step", "value": patches / dt, "unit": "patches/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32 (convolutions: f32 operands as 3 exact bf16 pieces on the bf16 MFMA, 6 of 9 piece products, f32 "
                 "accumulate, per-product worst case 2^-21 -- conv1's tap loops of the per-sample kernels as 2 fp16 pieces with "
                 "power-of-two operand scales, 3 piece products, worst case 2^-20 / mean 2^-23, falling back to the 3-piece loop "
                 "outside fp16's range; max-norm error measured at f32 level against f64; everything else f32 MFMA / "
                 "f32 VALU)", "data": "synthetic",
        "config": {"workload": f"{args.workload}: synthetic PaviaU-shaped patches {shape[1]}x{shape[2]}x{shape[0]}, "
                               f"spectrum {shape[3]}, {shape[4]} classes, {bt} labelled + {btu} unlabelled "
                               f"rows per GPU (batch {n_local}), dual BaseNet2 fwd/bwd + contrastive/mutual losses + "
                               f"bank + Adam, epoch 1 (memory-bank smoothing active), in-kernel PCG4D noise / Philox dropout",
                   "global_batch": n_local * world, "parallelism": f"dp{world}",
                   # how a rank's step is driven at N > 1: one C call with the collectives on RCCL directly, or the stages
                   # from Python over torch.distributed (the fallback of cmlpl_amd.distributed.pick_comm)
                   **({"collectives": type(eng.comm).__name__ + (", one C call per step" if eng._native_comm() is not False else ", stages driven from Python")}
                      if dist is not None else {})},
        "roofline": {"bound": "mfma", "kernel": labels[dom_name],
                     "achieved": achieved, "peak": kpeak[dom_name], "unit": "TFLOP/s",
                     "frac": achieved / kpeak[dom_name],
                     "peak_note": "fp32-equivalent MFMA ceiling of this kernel's own instruction mix: algorithmic "
                                  "FLOPs / (f32-segment FLOPs / 157.3 T + split-segment FLOPs / (2500 T / 6) + "
                                  "split2-segment FLOPs / (2500 T / 3)); a three-piece bf16 product costs six 16-bit "
                                  "MFMAs, a two-piece fp16 product three (history of the mix: DESIGN.md section 4)",
                     "flops_by_mfma_kind": kseg[dom_name],
                     "frac_of_f32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                     # continuity with rounds 2-5, whose kernels ran every product on three bf16 pieces: the same achieved
                     # rate against THAT mix's ceiling (2500 T / 6) -- `frac` above is against this kernel's own, higher one
                     "frac_of_three_piece_ceiling": achieved / SPLIT_PEAK_TFLOPS,
                     "traffic": traffic.get(dom_name),
                     "traffic_unit": "bytes/launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE; recorded in "
                                     f"{traffic_src or 'profiles/'}, null when the kernel sources changed since)",
                     "flops_per_launch": flops, "ms_per_launch": dom_ms, "launches_timed": int(cnt[dom_id])},
        # the other two conv1 kernels (same algorithmic FLOPs), from the calibration window
        "roofline_others": [
            {"kernel": labels[k], "ms_per_launch": calib[k], "launches_timed": calib_steps,
             "flops_per_launch": kflops[k], "flops_by_mfma_kind": kseg[k], "peak": kpeak[k],
             "achieved": kflops[k] / (calib[k] * 1e-3) / 1e12 if calib[k] > 0 else 0.0,
             "frac": (kflops[k] / (calib[k] * 1e-3) / 1e12 / kpeak[k]) if calib[k] > 0 else 0.0,
             "traffic": traffic.get(k)}
            for k in CONV1_KERNELS if k != dom_name],
        "final_losses": {k: scal[k] for k in ("total_s", "total_w", "cls_s", "ctr_s", "con_s")},
        # A sample whose gradient image is zero everywhere -- an unlabelled row no loss term reaches through the
        # convolutions, i.e. one under the confidence threshold (train.py:221: 0.999 in the first epochs) -- skips its
        # data-gradient loops and conv0 weight gradient in the fused backward and is left out of the weight-gradient
        # launch (exact: zeros times finite operands).  How many of the last step's unlabelled rows were ABOVE the
        # threshold, per network: the others were skipped.  CMLPL_ZERO_SKIP=0 measures with nothing skipped; the
        # algorithmic FLOPs of the backward / weight-gradient lines in roofline_others count the skipped rows as done.
        "backward_zero_images": {"unlabelled_rows": btu * world, "confident_s": scal["n_mask_s"], "confident_w": scal["n_mask_w"]},
        # did the container's CPU quota freeze this process inside the timed region?  (a throttled period stops every
        # thread for up to a CFS period, ~100 ms: profiles/r06_stall_rootcause.txt)  null where cpu.stat is not readable
        "host_throttled": None if thr0 is None or thr1 is None else
                          {"periods": thr1[0] - thr0[0], "ms": (thr1[1] - thr0[1]) / 1e3},
    }

    if args.breakdown:     # every rank runs the extra steps (they contain collectives); rank 0 prints
        _lib.check("cmlpl_timing_begin", lib.cmlpl_timing_begin(0xFFFFFFFF, 40 * 24))
        run(40, args.warmup + calib_steps + args.steps)
        barrier()
        _lib.check("cmlpl_timing_end", lib.cmlpl_timing_end(ms, cnt))
        tot = sum(ms[i] for i in range(len(_lib.KERNEL_NAMES)))
        if rank == 0:
            print(f"per-kernel device time over 40 steps (hipEvent pairs; sum {tot / 40 * 1e3:.1f} us/step):",
                  file=sys.stderr)
            for i, nm in enumerate(_lib.KERNEL_NAMES):
                if cnt[i]:
                    print(f"  {nm:12s} {ms[i] / cnt[i] * 1e3:9.1f} us/launch  x{cnt[i] // 40}/step  "
                          f"{100 * ms[i] / tot:5.1f} %", file=sys.stderr)

    if dist is not None:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        # on rank 0 at N = 1 only; a bounded sample (about 20 s) of the same workload on the host cores
        out["cpu_baseline"] = cpu_baseline(shape, bt, btu) if (world == 1 and not args.no_cpu_baseline) else None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


def cgroup_throttle():
    """(throttled periods, throttled time in us) of this container's CPU controller (cgroup v2, else v1), or None"""
    for path, key in (("/sys/fs/cgroup/cpu.stat", "throttled_usec"), ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time")):
        try:
            d = dict(ln.split() for ln in open(path).read().splitlines() if len(ln.split()) == 2)
            t = int(d.get(key, 0))
            return int(d.get("nr_throttled", 0)), (t // 1000 if key == "throttled_time" else t)
        except (OSError, ValueError):
            continue
    return None


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    from cmlpl_amd.launch import launched_by_rendezvous_env
    if not launched_by_rendezvous_env() and (args.gpus > 1 or os.environ.get("CMLPL_BENCH_SPAWN")):
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
