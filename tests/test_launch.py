"""The self-launcher behind `python bench.py --gpus N` (cmlpl_amd/launch.py), on CPU with gloo, world 2:
rendezvous environment, rank 0's stdout relayed, a failing rank fails the job; plus bench.py's batch
sharding arithmetic (BASELINE configs B3 / B5)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_launch_child.py")


def test_spawn_two_gloo_ranks_and_relay_rank0():
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(2, [sys.executable, CHILD, "ok"], timeout=300)
    assert rc == 0, out
    recs = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    assert recs == [{"sum": 3.0, "world": 2, "local_rank": 0}]
    assert "banner from rank 1" not in out


def test_failing_rank_fails_the_job():
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(2, [sys.executable, CHILD, "fail"], timeout=300)
    assert rc == 3


def test_bench_sharding_arithmetic():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gpus", "8", "--workload", "B3"])
    assert bench.per_rank_batch(a, 8) == (64, 64, "strong")                 # BASELINE configs[2]
    a = bench.parse_args(["--gpus", "8", "--workload", "B5", "--global-batch", "64+512"])
    assert bench.per_rank_batch(a, 8) == (8, 64, "strong")                  # BASELINE configs[4]
    a = bench.parse_args(["--gpus", "4"])
    assert bench.per_rank_batch(a, 4) == (128, 128, "weak")
    a = bench.parse_args(["--gpus", "3", "--global-batch", "64+512"])
    with pytest.raises(SystemExit):
        bench.per_rank_batch(a, 3)


def test_bench_parent_of_n_ranks_never_imports_torch_cuda(monkeypatch):
    """`python bench.py --gpus 2` in a process without rendezvous variables takes the launcher branch (and would
    start 2 children); with them it runs as a rank."""
    from cmlpl_amd import launch
    assert not launch.launched_by_rendezvous_env({})
    assert launch.launched_by_rendezvous_env({"RANK": "1", "WORLD_SIZE": "2"})
    env = launch.rank_env(1, 2, 12345, base={})
    assert env["RANK"] == "1" and env["LOCAL_RANK"] == "1" and env["WORLD_SIZE"] == "2"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "12345"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
