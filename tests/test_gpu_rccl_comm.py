"""GPU: the sharded step's collectives straight on RCCL (cmlpl_amd/rccl_comm.py, opt-in CMLPL_DIST_COMM=rccl) against the
torch.distributed path, one rank (this pool has one GPU per box and RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_comm_equals_torch_distributed_at_world_size_one():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_child.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "OK rccl" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
