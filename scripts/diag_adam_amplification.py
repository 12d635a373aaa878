import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import cmlpl_oracle as O
from tests.golden_util import GoldenCase
from tests.gpu_util import DEV, cuda_batch, hip_relu_gates, to_hp, to_shape
from cmlpl_amd import TrainEngine
g = GoldenCase("b2_64")
eng = TrainEngine(to_shape(g.shape), g.bt, g.btu, to_hp(g.hp), device=DEV)
p0, p1 = g.params(); eng.load_state_dict(0, p0); eng.load_state_dict(1, p1)
st = O.StepState.create(g.shape, p0, p1, g.bt, g.hp)
n = g.bt + g.btu
for s in range(2):
    b = g.batch(s); epoch, bi = g.epoch_bi(s); cb = cuda_batch(b)
    eng.step(cb["XPl"], cb["Xl"], cb["Y"], cb["XPu"], cb["Xu"], epoch, bi, noise=cb["noise"], dropmask=cb["dropmask"])
    gates = hip_relu_gates(eng, g.shape, n)
    ref = O.train_step(st, b["XPl"], b["Xl"], b["Y"], b["XPu"], b["Xu"], b["noise"], b["dropmask"], epoch, bi, g.hp, relu_gates=gates)
    gw = eng.grad(0, "feat_spe.weight").cpu().double(); rw = ref["grads"][0]["feat_spe.weight"].double()
    print("step", s, "lr", g.hp.lr)
    for col in (365, 100):
        print(" col", col, "|g| ref max", float(rw[col].abs().max()), "abs err max", float((gw[col]-rw[col]).abs().max()))
    lo, fe = eng.outputs()
    fr = torch.stack(ref["feats"]).double(); fe = fe.cpu().double()
    err = (fe - fr).abs()
    idx = torch.nonzero(err > 3e-6)
    print(" feat bad idx", idx[:12].tolist())
    y = eng.debug_region("y").view(2, n, 1024).cpu().double()
    zy = ref["taps"][0]["zy"].double()
    e = (y[0] - zy.clamp(min=0)).abs()
    print(" y err max", float(e.max()), "at", np.unravel_index(int(e.argmax()), e.shape), "norm y[30]", float(y[0,30].norm()))
    sd = eng.state_dict(0)["feat_spe.weight"].cpu().double(); sr = st.params[0]["feat_spe.weight"].double()
    d = (sd - sr).abs()
    print(" param err max", float(d.max()), "row of max", int(d.max(1).values.argmax()), "row365 sum|d|", float(d[365].sum()), "row365 max", float(d[365].max()))
