"""Micro-benchmark (GPU box): one NeRF training step (forward, loss, backward - no optimiser) at MI_RAYS rays (default 8192;
1024 = bench.py's nerf_train batch), for kernel-level profiles (rocprofv3 --kernel-trace) and A/B runs of two libraries
(MI_DIAG_LIB=gpurun_tools/<lib>.so) inside one gpurun call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import torch
from mirender import _lib, fields, render_core, train
if os.environ.get("MI_DIAG_LIB"):       # a diagnostic build of the library (tools/diag_build.sh), this tool only
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["MI_DIAG_LIB"])
    print("using", _lib.LIB_PATH, flush=True)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cm, fm = fields.NeRF().to(dev), fields.NeRF().to(dev)
n = int(os.environ.get("MI_RAYS", "8192"))
rays = torch.randn(n, 2, 3, device=dev); rays[:, 0] = torch.tensor([0., 0., 4.], device=dev); rays[:, 1, 2] = -1
tgt = torch.rand(n, 4, device=dev)
params = list(cm.parameters()) + list(fm.parameters())
def step():
    out = render_core.render_rays(rays, 2.0, 6.0, cm, fm, 64, 128)
    loss, _ = train.nerf_loss(out, tgt[:, :3], tgt[:, 3], use_alpha=True)
    for p in params: p.grad = None
    loss.backward()
for _ in range(2 if n > 2048 else 10): step()
torch.cuda.synchronize(); t = time.time(); k = 6 if n > 2048 else 100
for _ in range(k): step()
torch.cuda.synchronize(); dt = (time.time() - t) / k
print(f"nerf train n={n}: {dt*1e3:.2f} ms/step", flush=True)
