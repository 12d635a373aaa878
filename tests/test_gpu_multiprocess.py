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


def test_multiprocess_shards_by_index_equal_single_process():
    """the same job with every rank holding the whole (shuffled) splits and taking ITS rows through index lists
    (DistTrainEngine.step(lab_idx=, unl_idx=): what train.py does under torch.distributed.run)"""
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_child.py")],
                          extra_env=dict(CMLPL_TEST_BYIDX="1"), timeout=600)
    assert rc == 0, out
    assert "OK world=2" in out, out


def test_multiprocess_at_the_shard_size_of_configs2():
    """BASELINE configs[2] puts 64 + 64 rows on each of 8 ranks.  A GPU box of this pool admits at most six processes on
    its card at once (this test process is one of them), so the REAL wiring -- separate processes, rendezvous,
    DistTrainEngine + TorchDistComm, collectives between the stages -- runs here at that per-rank shard size with
    FOUR ranks (global 256 + 256); eight ranks are covered in lockstep in tests/test_gpu_distributed.py."""
    from cmlpl_amd.launch import spawn_ranks
    env = dict(CMLPL_TEST_BT="256", CMLPL_TEST_BTU="256", CMLPL_TEST_STEPS="2")
    rc, out = spawn_ranks(4, [sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_child.py")], extra_env=env,
                          timeout=900)
    assert rc == 0, out
    assert "OK world=4" in out, out


def test_multiprocess_replayed_stage_graphs_equal_the_eager_sharded_step():
    """DistStepGraph.launch() -- the seven stage graphs with the real torch.distributed calls between them, the embedding
    all-gather and the reduce-scatter issued asynchronously -- in two REAL processes, bit-identical to the eager
    drive_step on the same process group (what `train.py --graph` runs under torch.distributed.run)."""
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_child.py")],
                          extra_env=dict(CMLPL_TEST_GRAPH="1", CMLPL_TEST_BT="32", CMLPL_TEST_BTU="48"), timeout=600)
    assert rc == 0, out
    assert "OK graph world=2" in out, out
