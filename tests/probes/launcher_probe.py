"""Rank process started by bench.spawn_ranks in tests/test_launcher.py: joins a gloo group from the RANK / WORLD_SIZE /
MASTER_* environment the launcher sets, all-reduces, and rank 0 prints one JSON line (what bench.py's ranks do with RCCL)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if len(sys.argv) > 1 and sys.argv[1] == "--crash-early":      # rank 1 dies before it joins the group: rank 0 would wait in the rendezvous for minutes
    if rank == 1:
        sys.stderr.write("probe rank 1: crashing before the rendezvous\n")
        sys.exit(7)
    import time
    time.sleep(600)
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
dist.barrier()
if rank == 0:
    print(json.dumps({"world": world, "sum": float(t), "argv": sys.argv[1:]}))
dist.destroy_process_group()
sys.exit(3 if (len(sys.argv) > 1 and sys.argv[1] == "--fail" and rank == 1) else 0)
