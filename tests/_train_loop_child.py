"""Child program of tests/test_train_loop_gloo.py: one gloo rank running train.py's loop on CPU around the
test-only stand-in engine (tests/cpu_dist_engine.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import train  # noqa: E402
from cmlpl_amd.distributed import TorchDistComm  # noqa: E402
from tests.cpu_dist_engine import CpuLoopEngine  # noqa: E402

torch.set_num_threads(2)
out, num_unlabel = sys.argv[1], sys.argv[2]
dist.init_process_group("gloo")
comm = TorchDistComm()
args = train.build_parser().parse_args([
    "--synthetic", "B2", "--num_unlabel", num_unlabel, "--labeled_batch_size", "8", "--unlabeled_batch_size", "8",
    "--num_epochs", "2", "--print_per_batches", "2", "--no_eval", "--dropout", "0"])
hist = train.main(args, make_engine=lambda shape, bt_l, btu_l, hp, ppb: CpuLoopEngine(shape, bt_l, btu_l, hp, comm, ppb),
                  device=torch.device("cpu"))
import numpy as np  # noqa: E402
np.save(f"{out}_rank{comm.rank}.npy", hist)
dist.barrier()
dist.destroy_process_group()
