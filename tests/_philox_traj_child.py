"""Child of tests/test_gpu_env_paths.py: a few Philox-mode steps at B2 (64+64, or argv: SHAPE bt btu), printing every
scalar and a checksum of parameters / banks as hex floats, so that two kernel selections can be compared bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cmlpl_amd import HyperParams, NetShape, TrainEngine  # noqa: E402

dev = "cuda:0"
SHAPES = {"B2": (103, 11, 11, 103, 9), "B4": (200, 11, 11, 200, 16), "W8": (40, 8, 8, 40, 5), "W10": (64, 10, 11, 64, 33),
          "P": (60, 20, 20, 60, 16), "W12": (5, 12, 12, 9, 3)}
shp = SHAPES[sys.argv[1]] if len(sys.argv) > 1 else SHAPES["B2"]
bt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
btu = int(sys.argv[3]) if len(sys.argv) > 3 else 64
C_, H_, W_, bands_, K_ = shp
eng = TrainEngine(NetShape(*shp), bt, btu, HyperParams(), device=dev, seed=321)
eng.init_params_default(7)
g = torch.Generator().manual_seed(5)
b = [torch.randn(bt, C_, H_, W_, generator=g).to(dev), torch.randn(bt, bands_, generator=g).to(dev),
     torch.randint(0, K_, (bt,), generator=g).to(dev), torch.randn(btu, C_, H_, W_, generator=g).to(dev),
     torch.randn(btu, bands_, generator=g).to(dev)]
for s in range(3):
    eng.step(*b, 1, s)
    print(" ".join(float(v).hex() for v in eng.scalars.tolist()))
print(float(eng.params.double().sum()).hex(), float(eng.bank_feats.double().sum()).hex(),
      float(eng.grads.double().abs().sum()).hex())
import hashlib  # noqa: E402
h = hashlib.sha256()
for t in (eng.params, eng.m, eng.v, eng.grads, eng.bank_feats, eng.bank_probs, eng.logits, eng.feat):
    h.update(t.cpu().numpy().tobytes())
print("sha", h.hexdigest())
