import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import torch
from mirender import fields, render_core
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cm, fm = fields.NeRF().to(dev), fields.NeRF().to(dev)
n = 8192
rays = torch.randn(n, 2, 3, device=dev); rays[:, 0] = torch.tensor([0., 0., 4.], device=dev); rays[:, 1, 2] = -1
tgt = torch.rand(n, 3, device=dev)
params = list(cm.parameters()) + list(fm.parameters())
def fwd():
    out = render_core.render_rays(rays, 2.0, 6.0, cm, fm, 64, 128)
    return ((out[3] - tgt) ** 2).mean() + ((out[0] - tgt) ** 2).mean()
for _ in range(2): fwd().backward()
tf = tb = 0.0
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.time(); l = fwd(); torch.cuda.synchronize(); t1 = time.time()
    l.backward(); torch.cuda.synchronize(); t2 = time.time()
    tf += t1 - t0; tb += t2 - t1
print(f"MI_DBG={os.environ.get('MI_DBG')} fwd {tf/4*1e3:.2f} ms  bwd {tb/4*1e3:.2f} ms", flush=True)
