"""One-rank RCCL group on this GPU: the collectives bench.py uses between ranks (barrier, MAX all-reduce and all-gather of
float64 vectors on the device) run through the real backend.  A plumbing check for boxes with a single GPU; the 2-rank
logic is covered on CPU by tests/test_dist_gloo.py."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29537")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
import torch
import torch.distributed as dist
from pyruhvro_amd import dist as rdist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
print(rdist.init_process_group("nccl", dev))
dist.barrier(device_ids=[0])
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
v = torch.arange(7, dtype=torch.float64, device=dev)
outs = [torch.zeros_like(v)]
dist.all_gather(outs, v)
torch.cuda.synchronize()
assert float(t.item()) == 1.25 and outs[0].tolist() == list(range(7))
dist.destroy_process_group()
print("rccl one-rank ok")
