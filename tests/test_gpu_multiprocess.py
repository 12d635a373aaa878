"""GPU: the data-parallel step as REAL separate processes.  This pool has one GPU per box and RCCL refuses two ranks
on one device, so the ranks share cuda:0 and the collectives run on gloo; everything else is the shipped path:
cmlpl_amd.launch.spawn_ranks -> torch.distributed rendezvous -> DistTrainEngine + TorchDistComm -> the HIP kernels on
row shards -> all-gather / reduce-scatter / all-reduce between the stages.  Rank 0 holds the result to the
single-process TrainEngine on the same global batch."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_multiprocess_sharded_step_equals_single_process(world):
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(world, [sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_child.py")], timeout=600)
    assert rc == 0, out
    assert f"OK world={world}" in out, out
