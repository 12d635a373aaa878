"""Debug aid: per-tensor gradient error vs the oracle + run-to-run determinism of the HIP step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cmlpl_oracle as O
from tests.golden_util import GoldenCase
from tests.gpu_util import DEV, cuda_batch, to_hp, to_shape
from cmlpl_amd import TrainEngine

name = sys.argv[1] if len(sys.argv) > 1 else "p_traj_32"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
g = GoldenCase(name)
eng = TrainEngine(to_shape(g.shape), g.bt, g.btu, to_hp(g.hp), device=DEV)
p0, p1 = g.params()
eng.load_state_dict(0, p0); eng.load_state_dict(1, p1)
st = O.StepState.create(g.shape, p0, p1, g.bt, g.hp)
for s in range(min(nsteps, g.steps)):
    b = g.batch(s); epoch, bi = g.epoch_bi(s)
    ref = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"], epoch, bi, g.hp)
    cb = cuda_batch(b)
    # run 3x without update (banks are rewritten identically; ptr restored) to test determinism
    runs = []
    for rep in range(3):
        ptr = list(eng.ptr); sc = eng.step_count
        eng.step(cb["XPl"], cb["Xl"], cb["Y"], cb["XPu"], cb["Xu"], epoch, bi, noise=cb["noise"],
                 dropmask=cb["dropmask"], apply_update=False)
        torch.cuda.synchronize()
        runs.append(eng.grads.clone())
        eng.ptr = ptr; eng.step_count = sc
    det = [float((runs[0] - r).abs().max()) for r in runs[1:]]
    eng.step(cb["XPl"], cb["Xl"], cb["Y"], cb["XPu"], cb["Xu"], epoch, bi, noise=cb["noise"], dropmask=cb["dropmask"])
    torch.cuda.synchronize()
    line = []
    for net in range(2):
        for k in O.LIVE_KEYS:
            gr = ref["grads"][net][k]
            err = float((eng.grad(net, k).cpu() - gr).abs().max()) / max(float(gr.abs().max()), 1e-12)
            line.append(f"{err:.1e}")
    print(f"step {s} det={det} relerr(net0 then net1; {','.join(k.split('.')[0][:5]+k.split('.')[1][0] for k in O.LIVE_KEYS)}): {' '.join(line)}", flush=True)
