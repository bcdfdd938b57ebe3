"""Micro-benchmark (GPU box): the three MLP passes of a training step in isolation, per field kind - plain forward,
saving forward, backward (chain + dW GEMMs + reductions) - on one C4-sized range (4 images x 16 384 rays x 36 samples =
2.36 M points) for the FiLM field and on a 8 192-ray x 192-sample pass for NeRF / SirenNeRF.  TFLOP/s and % of the fp32
MFMA peak per pass (backward = 2x the forward's FLOPs).  MI_DIAG_LIB=gpurun_tools/<lib>.so runs a diagnostic build
(tools/diag_build.sh) for A/B comparisons.  Not a test; the judged numbers come from bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import torch
from mirender import _lib, autograd as A, fields, ops
if os.environ.get("MI_DIAG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["MI_DIAG_LIB"])
    print("using", _lib.LIB_PATH, flush=True)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
PEAK = 157.3


def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps


def run(name, cls, n, s, near, far, groups=0):
    m = cls().to(dev)
    pf = fields.as_packed_field(m)
    rays = torch.randn(n, 2, 3, device=dev); rays[:, 0] = torch.tensor([0., 0., 1.0 if groups else 4.], device=dev); rays[:, 1, 2] = -1
    z = torch.sort(torch.rand(n, s, device=dev) * (far - near) + near, -1).values
    film = (torch.rand((groups, 9, 512), device=dev) + 0.5) if groups else None
    fl = fields.FLOPS_PER_POINT[pf.kind] * n * s
    t_plain = timed(lambda: ops.field_eval_rays(pf, rays, z, film))
    raw, _acts = A._forward_saving(pf, rays, z, film)
    del _acts
    t_save = timed(lambda: A._forward_saving(pf, rays, z, film))
    g_raw = torch.randn_like(raw)
    # nothing kept: _field_backward re-runs the saving forward of the range, then the chain, the dW GEMMs and the reductions
    t_bwd = timed(lambda: A._field_backward(pf, rays, z, raw, g_raw, film, None))
    t_bwd -= t_save
    f = lambda t, k: f"{t * 1e3:8.2f} ms {k * fl / t / 1e12:6.1f} TFLOP/s ({k * fl / t / 1e12 / PEAK * 100:4.1f} %)"
    print(f"{name:6s} P={n * s:8d}  plain fwd {f(t_plain, 1)} | saving fwd {f(t_save, 1)} | backward {f(t_bwd, 2)}", flush=True)


run("film", fields.FilmSirenNeRF, 4 * 16384, 36, 0.5, 1.5, groups=4)
run("nerf", fields.NeRF, 8192, 192, 2.0, 6.0)
run("siren", fields.SirenNeRF, 8192, 192, 2.0, 6.0)


def pigan_inference():
    """pi_GAN inference (Generator.forward under no_grad: 32 images 128x128, 12+24, one FiLM field for both passes): the
    one-field path (the fine launch evaluates the 24 new depths only) against the two-field path on the same parameters
    (a second PackedField view: every point evaluated, 48 per ray, as the reference does)."""
    m = fields.FilmSirenNeRF().to(dev)
    pf = fields.as_packed_field(m)
    pf2 = fields.PackedField(pf.kind, pf.params)
    n = 32 * 128 * 128
    rays = torch.randn(n, 2, 3, device=dev); rays[:, 0] = torch.tensor([0., 0., 1.], device=dev); rays[:, 1, 2] = -1
    film = torch.rand((32, 9, 512), device=dev) + 0.5
    with torch.no_grad():
        t1 = timed(lambda: ops.render_rays_fused(pf, pf, rays, 0.5, 1.5, 12, 24, film, None, 1))
        t2 = timed(lambda: ops.render_rays_fused(pf, pf2, rays, 0.5, 1.5, 12, 24, film, None, 1))
    print(f"pi_GAN inference, 32 x 128x128, 12+24: one-field path {t1 * 1e3:.1f} ms ({n / t1 / 1e6:.2f} M rays/s, 36 evaluations per ray) | "
          f"every point evaluated {t2 * 1e3:.1f} ms ({n / t2 / 1e6:.2f} M rays/s, 48 per ray)", flush=True)


pigan_inference()
