"""Drop-in for the reference's ``tools/models.py`` hot-path surface: ``from tools.models import *``
keeps working (train.py:5).  ``BaseNet2`` / ``Normalize`` run on the gfx950 kernels of cmlpl_amd.
The reference's other classes in this file (ContrastiveLoss, CCT_Net, ...) are outside the hot path
 and are not provided."""
from cmlpl_amd.models import BaseNet2, Normalize  # noqa: F401

__all__ = ["BaseNet2", "Normalize"]
