"""Child of tests/test_gpu_env_paths.py: a few Philox-mode steps at B2 (64+64, or argv: SHAPE bt btu), printing every
scalar and a checksum of parameters / banks as hex floats, so that two kernel selections can be compared bit for bit.
`run()` is the same thing as a function: tests/_env_paths_child.py walks several kernel selections in ONE process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHAPES = {"B2": (103, 11, 11, 103, 9), "B4": (200, 11, 11, 200, 16), "W8": (40, 8, 8, 40, 5), "W10": (64, 10, 11, 64, 33),
          "P": (60, 20, 20, 60, 16), "W12": (5, 12, 12, 9, 3)}


def run(shape="B2", bt=64, btu=64, dev="cuda:0"):
    """the printed lines: three rows of step scalars, the checksums, 'sha <digest>' of every state tensor"""
    import hashlib

    import torch

    from cmlpl_amd import HyperParams, NetShape, TrainEngine
    shp = SHAPES[shape]
    C_, H_, W_, bands_, K_ = shp
    eng = TrainEngine(NetShape(*shp), bt, btu, HyperParams(), device=dev, seed=321)
    eng.init_params_default(7)
    g = torch.Generator().manual_seed(5)
    b = [torch.randn(bt, C_, H_, W_, generator=g).to(dev), torch.randn(bt, bands_, generator=g).to(dev),
         torch.randint(0, K_, (bt,), generator=g).to(dev), torch.randn(btu, C_, H_, W_, generator=g).to(dev),
         torch.randn(btu, bands_, generator=g).to(dev)]
    lines = []
    for s in range(3):
        eng.step(*b, 1, s)
        lines.append(" ".join(float(v).hex() for v in eng.scalars.tolist()))
    lines.append(" ".join((float(eng.params.double().sum()).hex(), float(eng.bank_feats.double().sum()).hex(),
                           float(eng.grads.double().abs().sum()).hex())))
    h = hashlib.sha256()
    for t in (eng.params, eng.m, eng.v, eng.grads, eng.bank_feats, eng.bank_probs, eng.logits, eng.feat):
        h.update(t.cpu().numpy().tobytes())
    lines.append("sha " + h.hexdigest())
    return lines


if __name__ == "__main__":
    a = sys.argv
    print("\n".join(run(a[1] if len(a) > 1 else "B2", int(a[2]) if len(a) > 2 else 64, int(a[3]) if len(a) > 3 else 64)))
