"""Child program of tests/test_launch.py: one rank of a gloo job started by cmlpl_amd.launch.spawn_ranks."""
import json
import os
import sys

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "fail" and rank == world - 1:
    sys.exit(3)
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print(f"banner from rank {rank}")          # non-zero ranks' stdout must not reach the parent's stdout
if rank == 0:
    print(json.dumps({"sum": float(t), "world": world, "local_rank": int(os.environ["LOCAL_RANK"])}))
dist.barrier()
dist.destroy_process_group()
