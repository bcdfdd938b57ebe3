"""RCCL smoke test on one GPU (1-rank process group): render_image_dist + allreduce_grads through the nccl backend."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
from mirender import dist as mdist
x = torch.arange(10, dtype=torch.float32, device="cuda").reshape(5, 2)
recv = torch.empty_like(x)
dist.all_gather_into_tensor(recv, x)
t = torch.ones(3, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
print("nccl ok", recv.sum().item(), t.tolist(), mdist.all_gather_rays(x, 5).shape)
p = [torch.nn.Parameter(torch.zeros(4, device="cuda"))]; p[0].grad = torch.ones(4, device="cuda"); mdist.allreduce_grads(p)
dist.destroy_process_group()
print("done")
