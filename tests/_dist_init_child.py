"""Child of tests/test_dist_startup.py: one rank of a CPU (gloo) job that goes through init_distributed and the
debug-mode collectives of TorchDistComm."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cmlpl_amd.distributed import TorchDistComm, init_distributed, pick_comm  # noqa: E402

mode = sys.argv[1]
if mode == "hang":                       # rank 1 never joins: rank 0's watchdog must end it with code 3
    if os.environ["RANK"] == "1":
        import time
        time.sleep(60)
        sys.exit(0)
    init_distributed("gloo", None, timeout_s=4.0)
    sys.exit(0)
dist = init_distributed("gloo", None, timeout_s=60.0)
assert isinstance(pick_comm("cpu"), TorchDistComm)       # a gloo job: never the direct RCCL communicator
comm = TorchDistComm(debug=True)
W, r = comm.world, comm.rank
inp = torch.arange(6, dtype=torch.float32) + 100 * r
out = torch.zeros(6 * W)
comm.all_gather(out, inp)
assert torch.equal(out.view(W, 6)[r], inp)
t = torch.full((5,), float(r + 1))
comm.all_reduce(t)
assert torch.equal(t, torch.full((5,), W * (W + 1) / 2))
rs_in = torch.arange(4 * W, dtype=torch.float32) * (r + 1)
rs_out = torch.zeros(4)
comm.reduce_scatter(rs_out, rs_in)
want = torch.arange(4 * W, dtype=torch.float32).view(W, 4)[r] * (W * (W + 1) / 2)
assert torch.equal(rs_out, want), (rs_out, want)
dist.destroy_process_group()
if r == 0:
    print("startup ok")
