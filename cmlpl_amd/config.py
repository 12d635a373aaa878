"""Shape and hyper-parameter records of the CMLPL hot path (host side)."""
from __future__ import annotations

import math
from dataclasses import dataclass

FEAT_DIM = 1024   # tools/models.py:119 of the reference
CONV_CH = 64


@dataclass(frozen=True)
class NetShape:
    """One BaseNet2: conv0 in-channels C, window HxW, spectrum length, classes.
    Reference literals: C=60, 20x20 (classifier 2624 wide), tools/models.py:102,127."""
    C: int = 60
    H: int = 20
    W: int = 20
    bands: int = 103
    K: int = 9

    @property
    def cls_in(self) -> int:
        return CONV_CH * ((self.H // 2) // 2) * ((self.W // 2) // 2) + FEAT_DIM


@dataclass
class HyperParams:
    """train.py:356-379 flags + the literals of the step (same names as the flags)."""
    lr: float = 5e-4
    num_epochs: int = 20
    thr: float = 1.0
    alpha: float = 0.95
    queue_batch: float = 17
    temperature: float = 0.3
    dropout: float = 0.8
    noise: float = 0.5
    w_contrast: float = 0.5      # train.py:266,270
    w_mutual: float = 4.0
    pos_thr: float = 0.8         # train.py:251
    neg_thr: float = 0.3         # train.py:254
    bank_step: int = 256         # literal pointer advance, train.py:234,237
    bank_mult: int = 5           # queue_size = 5 * labeled_batch_size * 2, train.py:138
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8

    def adap_thr(self, epoch: int) -> float:
        """train.py:147-148"""
        return math.exp(-0.5 * ((epoch / self.num_epochs) ** 2))

    def smooth_gate(self, epoch: int, batch_index: int) -> bool:
        """train.py:212"""
        return epoch > 0 or batch_index > self.queue_batch
