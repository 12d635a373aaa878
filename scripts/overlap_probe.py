"""Do the step's latency-bound loss launches overlap with the big per-sample launches when they run on a second stream?

Timing probe only (results of the overlapped variants are meaningless: the side stream reads the PREVIOUS step's
features / logits).  The stages of the sharded step at world size 1 are separate C calls that take a stream, so they can
be placed freely:
  seq            the stages of the sharded step on one stream (= the product's order)
  fwd||p1        phase 1 (pair_exp + loss_rows) on stream B beside the forward (spe_fused + fused forward) on stream A
  bwd||p2        phase 2 (graph_loss + loss_dfeat) on stream B beside the backward on stream A
  both
for B at normal and at low priority.  What north_star's dependency structure allows in the real step: pair_exp needs the
embeddings only (spectral branch), the convolution backward needs dlogits only (tools/models.py:142-150).
    python scripts/overlap_probe.py [B2 128 128]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cmlpl_amd import NetShape, HyperParams
from cmlpl_amd.distributed import DistTrainEngine
from bench import synth, WORKLOADS

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "B2"
bt = int(sys.argv[2]) if len(sys.argv) > 2 else 128
btu = int(sys.argv[3]) if len(sys.argv) > 3 else 128
shape = WORKLOADS[wl]
b = synth(shape, bt, btu, 1, dev)
eng = DistTrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=dev, seed=1088)
eng.init_params_default(1088)
args = (b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"])
K = 300


def step(mode, sB, i):
    A = torch.cuda.current_stream(dev)
    if mode in ("fwd||p1", "both"):
        sB.wait_stream(A)
        eng.stage_spectral(*args, 1, i); eng.stage_spatial()
        with torch.cuda.stream(sB):
            eng.stage_phase1()
        A.wait_stream(sB)
    else:
        eng.stage_spectral(*args, 1, i); eng.stage_spatial()
        eng.stage_phase1()
    if mode in ("bwd||p2", "both"):
        sB.wait_stream(A)
        with torch.cuda.stream(sB):
            eng.stage_phase2()
        eng.stage_backward_data(); eng.stage_backward_weights()
        A.wait_stream(sB)
    else:
        eng.stage_phase2()
        eng.stage_backward_data(); eng.stage_backward_weights()
    eng.stage_update()


for i in range(30):
    step("seq", None, i)
torch.cuda.synchronize()
print(f"{wl} {bt}+{btu}: wall us/step over {K} steps")
for prio_name, prio in (("normal", 0), ("low", 1), ("high", -1)):
    sB = torch.cuda.Stream(device=dev, priority=prio)
    for mode in ("seq", "fwd||p1", "bwd||p2", "both"):
        for i in range(20):
            step(mode, sB, 100 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            step(mode, sB, 200 + i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"  side stream {prio_name:6s} {mode:8s} {1e6 * (t1 - t0) / K:7.1f}", flush=True)
