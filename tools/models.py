"""Drop-in for the reference's ``tools/models.py`` hot-path surface: ``from tools.models import *``
keeps working (train.py:5).  ``BaseNet2`` / ``Normalize`` run on the gfx950 kernels of cmlpl_amd.
``ContrastiveLoss`` (NT-Xent) is provided on the same kernels.  The reference's remaining
classes in this file (CCT_Net, Spa/SpeRandomization, ...) are outside the hot path (SURVEY.md section 2,
C11) and are not provided."""
from cmlpl_amd.models import BaseNet2, ContrastiveLoss, Normalize  # noqa: F401

__all__ = ["BaseNet2", "Normalize", "ContrastiveLoss"]
