"""Child of tests/test_gpu_env_paths.py: a few Philox-mode steps at B2 (64+64), printing every scalar and a
checksum of parameters / banks as hex floats, so that two kernel selections can be compared bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cmlpl_amd import HyperParams, NetShape, TrainEngine  # noqa: E402

dev = "cuda:0"
eng = TrainEngine(NetShape(103, 11, 11, 103, 9), 64, 64, HyperParams(), device=dev, seed=321)
eng.init_params_default(7)
g = torch.Generator().manual_seed(5)
b = [torch.randn(64, 103, 11, 11, generator=g).to(dev), torch.randn(64, 103, generator=g).to(dev),
     torch.randint(0, 9, (64,), generator=g).to(dev), torch.randn(64, 103, 11, 11, generator=g).to(dev),
     torch.randn(64, 103, generator=g).to(dev)]
for s in range(3):
    eng.step(*b, 1, s)
    print(" ".join(float(v).hex() for v in eng.scalars.tolist()))
print(float(eng.params.double().sum()).hex(), float(eng.bank_feats.double().sum()).hex(),
      float(eng.grads.double().abs().sum()).hex())
