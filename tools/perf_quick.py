"""Micro-benchmark (GPU box): fused field MLP in points mode for every kind, as % of the fp32 MFMA peak, and one
NeRF training step at 8192 rays.  Not a test; the judged numbers come from bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import numpy as np, torch
from mirender import _lib, fields, ops, render_core
if os.environ.get("MI_DIAG_LIB"):       # a diagnostic build of the library (tools/diag_build.sh), this tool only
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["MI_DIAG_LIB"])
    import ctypes
    _probe = ctypes.CDLL(_lib.LIB_PATH)       # a diagnostic build may predate the newest entry points
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
    print("using", _lib.LIB_PATH, flush=True)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
M = 1 << 22
x = torch.rand((M, 6), device=dev) * 2 - 1
for name, cls in (("nerf", fields.NeRF), ("tiny", fields.TinyNeRF), ("siren", fields.SirenNeRF), ("film", fields.FilmSirenNeRF)):
    m = cls().to(dev); pf = fields.as_packed_field(m)
    film = torch.rand((4, 9, 512), device=dev) + 0.5 if name == "film" else None
    fields.eval_points(pf, x, film); torch.cuda.synchronize()
    t = time.time(); k = 5
    for _ in range(k): fields.eval_points(pf, x, film)
    torch.cuda.synchronize(); dt = (time.time() - t) / k
    fl = fields.FLOPS_PER_POINT[pf.kind]
    print(f"{name:6s} fwd {M/dt/1e6:7.1f} Mpts/s  {M*fl/dt/1e12:6.1f} TFLOP/s ({M*fl/dt/1e12/157.3*100:.1f}% of fp32 MFMA peak)", flush=True)
if os.environ.get("MI_DIAG_LIB"):
    sys.exit(0)
cm, fm = fields.NeRF().to(dev), fields.NeRF().to(dev)
n = 8192
rays = torch.randn(n, 2, 3, device=dev); rays[:, 0] = torch.tensor([0., 0., 4.], device=dev); rays[:, 1, 2] = -1
tgt = torch.rand(n, 3, device=dev)
def step():
    out = render_core.render_rays(rays, 2.0, 6.0, cm, fm, 64, 128)
    loss = ((out[3] - tgt) ** 2).mean() + ((out[0] - tgt) ** 2).mean()
    for p in list(cm.parameters()) + list(fm.parameters()): p.grad = None
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize(); t = time.time(); k = 5
for _ in range(k): step()
torch.cuda.synchronize(); dt = (time.time() - t) / k
print(f"nerf train n={n}: {dt*1e3:.2f} ms/step  {n/dt:.0f} rays/s  algorithmic {n*256*3*2*591488/dt/1e12:.1f} TFLOP/s", flush=True)
