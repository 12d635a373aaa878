"""CPU: the checked start-up of a one-process-per-GPU job (cmlpl_amd.distributed.init_distributed) and the debug
mode of the collective wrappers, as two real processes over gloo; failure modes end with a message and a non-zero
code instead of a hang."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_dist_init_child.py")


def test_two_ranks_start_up_checked_and_debug_collectives_agree():
    from cmlpl_amd.launch import spawn_ranks
    rc, out = spawn_ranks(2, [sys.executable, CHILD, "ok"], timeout=300)
    assert rc == 0, out
    assert "startup ok" in out


def test_a_rank_that_never_joins_ends_the_job_instead_of_hanging():
    import time
    from cmlpl_amd.launch import spawn_ranks
    t0 = time.monotonic()
    rc, _ = spawn_ranks(2, [sys.executable, CHILD, "hang"], timeout=120, retries=0)
    assert rc != 0                                   # rank 0's watchdog (code 3) or the store's own timeout
    assert time.monotonic() - t0 < 60                # ... long before rank 1's sleep or the launcher's limit


def test_too_few_gpus_is_reported_before_any_collective(monkeypatch):
    import torch
    from cmlpl_amd.distributed import DistStartupError, init_distributed
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices the check asks for")
    with pytest.raises(DistStartupError, match="visible GPU"):
        init_distributed("nccl", torch.device("cuda:1"))
