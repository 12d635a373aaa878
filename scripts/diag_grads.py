"""Diagnostic: per-tensor error of one BaseNet2 forward/backward against the CPU oracle (no assertions).
   python scripts/diag_grads.py P 64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cmlpl_oracle as O
from tests.test_gpu_ops import SHAPES, _module
DEV = "cuda:0"
name, n = sys.argv[1], int(sys.argv[2])
shape = SHAPES[name]
params = O.closed_form_params(shape, 7)
g = torch.Generator().manual_seed(100 + n)
x = torch.randn(n, shape.C, shape.H, shape.W, generator=g)
y = torch.randn(n, shape.bands, generator=g)
keep = 0.2
dm = (torch.rand(n, shape.cls_in, generator=g) < keep).float() / keep
dlog = torch.randn(n, shape.K, generator=g)
dfe = torch.randn(n, 1024, generator=g) * 0.1
pr = {k: v.clone().double().requires_grad_(k in O.LIVE_KEYS) for k, v in params.items()}
lo_ref, fe_ref = O.basenet2_forward(pr, x.double(), y.double(), dm.double())
(lo_ref * dlog.double()).sum().add((fe_ref * dfe.double()).sum()).backward()
net = _module(shape, params, dropout=0.8)
net.train()
lo, fe = net(x.to(DEV), y.to(DEV), dropmask=dm.to(DEV))
((lo * dlog.to(DEV)).sum() + (fe * dfe.to(DEV)).sum()).backward()
torch.cuda.synchronize()
def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().double()
    return float((a - b).abs().max() / b.abs().max())
print(f"{name}-{n} (vs fp64 oracle)  logits {rel(lo, lo_ref):.2e}  feat {rel(fe, fe_ref):.2e}")
hip = dict(net.named_parameters())
for k in O.LIVE_KEYS:
    print(f"  grad {k:22s} max-rel err {rel(hip[k].grad, pr[k].grad):.2e}")
