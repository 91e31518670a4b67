import os, sys; sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from qtos_amd.dist import gather_plans
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
n = torch.arange(7 * 5, dtype=torch.float64, device="cuda").reshape(7, 5)
s = torch.arange(7, dtype=torch.int32, device="cuda")
a, b = gather_plans(n, s, 7)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl world 1:", torch.equal(a, n), torch.equal(b, s), float(t))
dist.destroy_process_group()
