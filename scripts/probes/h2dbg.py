import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cmlpl_amd import TrainEngine, NetShape, HyperParams, _lib
from bench import synth, WORKLOADS
lib = _lib.load()
dev = torch.device("cuda:0")
shape = WORKLOADS["B2"]
b = synth(shape, 128, 128, 1, dev)
res = {}
for X in ("0", sys.argv[1] if len(sys.argv) > 1 else "1"):
    os.environ["CMLPL_F16X2"] = X
    lib.cmlpl_debug_reload_switches()
    e = TrainEngine(NetShape(*shape), 128, 128, HyperParams(), device=dev, seed=1088)
    e.init_params_default(1088)
    e.step(b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], 1, 0, apply_update=False)
    torch.cuda.synchronize()
    res[X] = (e.logits.clone(), e.grads.clone())
    L = e.layout
for i, nm in enumerate(("logits", "grads")):
    a, c = res["0"][i], res[X][i]
    print(nm, "equal" if torch.equal(a, c) else f"differ: max abs {(a - c).abs().max().item():.3e} rel to max {((a - c).abs().max() / a.abs().max()).item():.3e}")
g0, g1 = res["0"][1], res[X][1]
off = [int(x) for x in L.param_off[:11]]
names = ["conv0.w", "conv0.b", "conv1.w", "conv1.b", "conv2.w", "conv2.b", "spe.w", "spe.b", "cls.w", "cls.b"]
for i, nm in enumerate(names):
    a, c = g0[:, off[i]:off[i + 1]], g1[:, off[i]:off[i + 1]]
    print(f"  {nm:8s} {'equal' if torch.equal(a, c) else 'differ %.3e / max %.3e' % ((a - c).abs().max().item(), a.abs().max().item())}")
