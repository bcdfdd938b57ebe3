"""Diagnostic: host-side (Python) time of one nerf training step, by function (cProfile), GPU running asynchronously."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import torch
from mirender import fields, render_core, train
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cm, fm = fields.NeRF().to(dev), fields.NeRF().to(dev)
n = 1024
rays = torch.randn(n, 2, 3, device=dev); rays[:, 0] = torch.tensor([0., 0., 4.], device=dev); rays[:, 1, 2] = -1
tgt = torch.rand(n, 4, device=dev)
opt = train.FusedAdam([cm, fm], lr=5e-4)
def step(i):
    out = render_core.render_rays(rays, 2.0, 6.0, cm, fm, 64, 128, seed=i)
    loss, _ = train.nerf_loss(out, tgt[:, :3], tgt[:, 3], use_alpha=True)
    opt.zero_grad()
    loss.backward()
    opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(20): step(i)
t_issue = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t) / 20
print(f"host issue time {t_issue*1e3:.2f} ms/step, wall {t_all*1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
