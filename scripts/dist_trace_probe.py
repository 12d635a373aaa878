"""Does the re-staged sharded step put its two large collectives UNDER the convolutions on the device?  One rank with the
four REAL torch.distributed calls (RCCL at world size 1: its collectives are device copies on RCCL's own stream), a few
dozen steps (CMLPL_DIST_COMM=rccl: the one-call step over RcclComm instead); run it under `rocprofv3 --kernel-trace` and
read the trace with `--summarise`:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 scripts/dist_trace_probe.py [B2 64 64]
    python3 scripts/dist_trace_probe.py --summarise OUT/.../t_kernel_trace.csv
The summary lists one steady-state step in start order: every kernel with its start / end relative to the step's first
kernel, and for every kernel that is not one of this library's (= a collective's copy) which library kernels it ran beside."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(wl, bt, btu, steps=40):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29518")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from cmlpl_amd import NetShape, HyperParams
    from cmlpl_amd.distributed import DistTrainEngine, TorchDistComm
    from bench import synth, WORKLOADS
    shape = WORKLOADS[wl]
    b = synth(shape, bt, btu, 1, dev)
    if os.environ.get("CMLPL_DIST_COMM") == "rccl":      # the one-call step (cmlpl_dist_step) with the collectives issued from C
        from cmlpl_amd.rccl_comm import RcclComm
        comm = RcclComm(dev)
    else:
        comm = TorchDistComm()
    eng = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088, comm=comm, alias_single=False)
    eng.init_params_default(1088)
    for i in range(steps):
        eng.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, i)
    torch.cuda.synchronize()
    dist.destroy_process_group()


def summarise(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    ours = lambda n: "cmlpl::" in n
    short = lambda n: n.replace("void ", "").replace("cmlpl::", "").split("(")[0][:44]
    # a step starts at a spe_fused launch; take the last complete one but two
    starts = [i for i, (_, _, n) in enumerate(rows) if "spe_fused" in n]
    if len(starts) < 4:
        print("fewer than four steps in the trace"); return
    i0, i1 = starts[-3], starts[-2]
    t0 = rows[i0][0]
    step = rows[i0:i1]
    print(f"one steady-state step ({len(step)} kernels, {(rows[i1][0] - t0) / 1e3:.1f} us from its first kernel to the next step's first):")
    for s, e, n in step:
        tag = "" if ours(n) else "   <-- not this library's: a collective's copy / torch"
        print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  {short(n)}{tag}")
    print("overlap of the foreign kernels with this library's:")
    for s, e, n in step:
        if ours(n):
            continue
        beside = [short(m) for (a, b, m) in step if ours(m) and a < e and b > s]
        print(f"  {short(n):44s} {(e - s) / 1e3:6.1f} us   beside: {', '.join(beside) if beside else '(nothing: exposed)'}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        a = sys.argv[1:]
        run(a[0] if a else "B2", int(a[1]) if len(a) > 1 else 64, int(a[2]) if len(a) > 2 else 64)
