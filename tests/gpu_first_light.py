"""First-light script for the GPU box: tiny calls, prints errors instead of asserting (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import numpy as np, torch
from mirender import fields, ops, _lib
from oracle import fields as ofields, render_ref as R, synth
dev = torch.device("cuda", 0)
print(torch.cuda.get_device_name(0), flush=True)
for kind_name in ("nerf", "tiny_nerf", "siren_nerf", "film_siren_nerf", "film_siren_nerf_nodir"):
    kind = {v: k for k, v in fields.KIND_NAMES.items()}[kind_name]
    sd = synth.state_dict(kind_name, seed=10, sharp=True, bias_jitter=0.05)
    params = []
    for key, _ in fields.SPECS[kind]:
        params += [sd[key + ".weight"].to(dev), sd[key + ".bias"].to(dev)]
    pf = fields.PackedField(kind, params)
    x = np.random.default_rng(0).uniform(-1.5, 1.5, size=(300, 6)).astype(np.float32)
    film = synth.film_params(1, 1)
    with torch.no_grad():
        ref = ofields.make_field(kind_name, sd, film[0])(torch.from_numpy(x)).numpy()
    out = fields.eval_points(pf, torch.from_numpy(x).to(dev), film.to(dev) if kind_name.startswith("film") else None)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    err = np.abs(out - ref)
    print(kind_name, "max err rgb", err[:, :3].max(), "sigma rel", (err[:, 3] / np.maximum(1, np.abs(ref[:, 3]))).max(),
          "sample", out[0], ref[0], flush=True)
# timing of the NeRF kernel
sd = synth.state_dict("nerf", seed=0, sharp=True)
params = []
for key, _ in fields.SPECS[0]:
    params += [sd[key + ".weight"].to(dev), sd[key + ".bias"].to(dev)]
pf = fields.PackedField(0, params)
for M in (1 << 15, 1 << 20, 1 << 22):
    x = torch.rand((M, 6), device=dev) * 2 - 1
    fields.eval_points(pf, x); torch.cuda.synchronize()
    t = time.time(); n = 3
    for _ in range(n): fields.eval_points(pf, x)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    print(f"nerf M={M}: {dt*1e3:.2f} ms  {M/dt/1e6:.1f} Mpts/s  {M*2*591488/dt/1e12:.1f} TFLOP/s", flush=True)
