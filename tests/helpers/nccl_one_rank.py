"""The RCCL ("nccl") branch of agarcl_amd.dist on ONE GPU: a process group of world size 1, the gatherers on CUDA tensors (TEST INFRASTRUCTURE; a child
process of tests/test_00_dist_gpu.py).  What a 1-GPU box can check of the path the driver's multi-GPU runs take: the calls are valid for the backend."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from agarcl_amd import dist as agdist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[1]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 37
g = agdist.ResultGatherer(n, dev, depth=2)
for k in range(4):
    rewards = torch.arange(n, dtype=torch.float64, device=dev) + 100 * k
    dones = (torch.arange(n, device=dev) % 3 == k % 3).to(torch.uint8)
    slot = g.pack(k, rewards, dones); g.wait_slot(slot)
    got = g.gathered(slot)
    assert torch.equal(got[:, 0], rewards.float()) and torch.equal(got[:, 1], dones.float())
    packed = torch.stack([rewards.float() * 2, dones.float()], dim=1).contiguous()
    g.gather_packed(k & 1, packed); g.wait_slot(k & 1)
    assert torch.equal(g.gathered(k & 1), packed)
tg = agdist.TensorGatherer((n, 6, 6, 3), torch.uint8, dev)
fr = torch.randint(0, 255, (n, 6, 6, 3), dtype=torch.uint8, device=dev)
tg.gather(fr); tg.wait()
assert torch.equal(tg.gathered(), fr)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 1.5
objs = [None]; dist.all_gather_object(objs, {"rank": 0}); assert objs[0]["rank"] == 0
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl one-rank ok")
