"""Not a test: frames per second of render_video (device->host copies overlapped with the next frame) against a loop of
render_image calls (three blocking pageable copies per frame), 800x800 @ 64 samples and 128x128 @ 12+24.
Usage (GPU box): python tools/perf_video.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "msra-practice-project_amd"))
sys.path.insert(0, ROOT)
from mirender import fields, render_core
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cases = [("nerf 800x800 64+0", fields.NeRF().to(dev), 800, 800, 64, 0, 2.0, 6.0, 8),
         ("nerf 200x200 64+128", fields.NeRF().to(dev), 200, 200, 64, 128, 2.0, 6.0, 16),
         ("tiny 100x100 32+0", fields.TinyNeRF().to(dev), 100, 100, 32, 0, 2.0, 6.0, 64)]
def pose(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, -s, -4 * s], [0, 1, 0, 0], [s, 0, c, 4 * c], [0, 0, 0, 1]], np.float32)
for name, m, W, H, nc, nf, near, far, F in cases:
    poses = [pose(a) for a in np.linspace(0, 6.28, F, endpoint=False)]
    render_core.render_video(W, H, 1.3875 * W, poses[:2], near, far, m, m, nc, nf, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, p in enumerate(poses):
        render_core.render_image(W, H, 1.3875 * W, p, near, far, m, m, nc, nf, seed=1 + i)
    t1 = time.perf_counter()
    render_core.render_video(W, H, 1.3875 * W, poses, near, far, m, m, nc, nf, seed=1)
    t2 = time.perf_counter()
    print(f"{name:24s} render_image loop {1e3 * (t1 - t0) / F:8.2f} ms/frame   render_video {1e3 * (t2 - t1) / F:8.2f} ms/frame")
