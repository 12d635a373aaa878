"""cmlpl_amd -- MI355X-native implementation of the CMLPL per-step training hot path.

Public surface:
  NetShape, HyperParams ........ shape / flag records
  TrainEngine .................. the fused training step (train.py:150-278 of the reference)
  BaseNet2 ..................... drop-in nn.Module (tools/models.py:97-152 of the reference)
The compute runs in libcmlpl_hip.so (hand-written gfx950 kernels, C ABI in include/cmlpl.h);
importing the package does not load it, using any op does -- and fails loudly if it is missing.
"""
from .config import HyperParams, NetShape  # noqa: F401


def __getattr__(name):
    if name == "TrainEngine":
        from .engine import TrainEngine
        return TrainEngine
    if name in ("BaseNet2", "Normalize"):
        from . import models
        return getattr(models, name)
    raise AttributeError(name)
