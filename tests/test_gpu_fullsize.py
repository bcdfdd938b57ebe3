"""GPU: BASELINE.json's configurations at their FULL sizes, through properties that need no oracle run (the oracle
takes ~10 minutes per 800x800 frame): invariance of the frame to how its rays are split over calls, determinism of the
seeded jitter, white background / range / finiteness, exact linearity of the backward in the cotangent, equality of
the training forward with the inference forward.  The small-size twins of these paths are checked against the oracle
link by link in test_gpu_render.py / test_gpu_train.py / test_gpu_c5.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import synth  # noqa: E402


def dev():
    return torch.device("cuda", 0)


def test_c3_full_frame_800x800_64_128():
    """C3: 800x800 = 640 000 rays, 64 + 128 samples, separate coarse / fine NeRF 8x256 - one frame in one launch
    sequence, the same frame as two ray ranges (what two GPUs render) and again with the same seed."""
    from mirender import fields, render_core
    W = H = 800
    cm = fields.field_from_state_dict(synth.state_dict("nerf", seed=0, sharp=True, bias_jitter=0.05), dev())
    fm = fields.field_from_state_dict(synth.state_dict("nerf", seed=1, sharp=True, bias_jitter=0.05), dev())
    pose, focal = synth.pose_degrees(4.0, 30.0, -30.0), 1.3875 * W
    with torch.no_grad():
        whole = render_core._render_image_device(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, None, None, 7, 0, W * H)
        cut = 800 * 311 + 97                                    # a split that is not on a row boundary
        a = render_core._render_image_device(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, None, None, 7, 0, cut)
        b = render_core._render_image_device(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, None, None, 7, cut, W * H - cut)
        again = render_core._render_image_device(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, None, None, 7, 0, W * H)
        other = render_core._render_image_device(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, None, None, 8, 0, 4096)
    for w, x, y, z in zip(whole, a, b, again):
        assert torch.equal(w, torch.cat([x, y])) and torch.equal(w, z)
    assert not torch.equal(whole[0][:4096], other[0])           # another seed is another jitter
    rgb, depth, acc = whole
    assert tuple(rgb.shape) == (W * H, 3) and bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    assert float(rgb.min()) >= 0 and float(rgb.max()) <= 1 + 1e-5 and float(acc.min()) >= 0 and float(acc.max()) <= 1 + 1e-5
    assert float(depth.max()) <= 6.0 * (1 + 1e-5) * float(acc.max()) + 1e-3
    # numpy surface: shapes and dtypes of nerf/render.py:165-167
    out = render_core.render_image(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, seed=7)
    assert [o.shape for o in out] == [(H, W, 3), (H, W, 1), (H, W, 1)] and all(o.dtype == np.float32 for o in out)
    assert np.array_equal(out[0].reshape(-1, 3), rgb.cpu().numpy())


def test_largest_single_launch_has_offsets_past_4_gib():
    """Maximum sizes: render_image hands the library at most 2^20 rays per launch (render_core.MAX_RAYS_PER_LAUNCH).
    At 64 + 128 samples such a launch holds 201 M fine points: raw_fine alone is 3.2 GB and the workspace 5.6 GB, so
    byte offsets inside one buffer pass 2^31 and 2^32 (an 800x800 frame stays under 2^31 everywhere).  A 1200x1000
    frame = one full 2^20-ray launch + a 151 424-ray remainder; its last rows (whose points sit at the far end of the
    big buffers) and the rows either side of the launch boundary must be bit-equal to the same rays rendered in small
    launches of their own."""
    from mirender import fields, render_core
    W, H = 1200, 1000
    assert W * H > render_core.MAX_RAYS_PER_LAUNCH == 1 << 20
    cm = fields.field_from_state_dict(synth.state_dict("nerf", seed=0, sharp="medium", bias_jitter=0.05), dev())
    fm = fields.field_from_state_dict(synth.state_dict("nerf", seed=1, sharp="medium", bias_jitter=0.05), dev())
    pose, focal = synth.pose_degrees(4.0, 30.0, -30.0), 1.3875 * W
    with torch.no_grad():
        whole = render_core._render_image_device(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, None, None, 5, 0, W * H)
        for r0, n in (((1 << 20) - 2048, 2048), ((1 << 20) - 700, 1400), (1 << 20, 1024), (W * H - 1024, 1024), (0, 512)):
            part = render_core._render_image_device(W, H, focal, pose, 2.0, 6.0, cm, fm, 64, 128, None, None, 5, r0, n)
            for w, x in zip(whole, part):
                assert torch.equal(w[r0:r0 + n], x), (r0, n)
    assert bool(torch.isfinite(whole[0]).all()) and float(whole[0].std()) > 1e-2
    torch.cuda.empty_cache()


def _generator(res, nc, nf):
    from mirender import pigan
    torch.manual_seed(0)
    gen = pigan.Generator(256, res, near=0.5, far=1.5, fov=12, coarse_samples=nc, fine_samples=nf).to(dev())
    gen.film_siren_nerf.load_state_dict(synth.state_dict("film_siren_nerf", seed=40, sharp="medium"))
    return gen


@pytest.mark.parametrize("name,res,batch,nc,nf", [("C4", 128, 32, 12, 24), ("C5 per GPU", 256, 4, 24, 48)])
def test_pigan_training_step_full_size(name, res, batch, nc, nf):
    """C4 (128x128, batch 32, 12+24) and one GPU's share of C5 (256x256, batch 4 of the global 32, 24+48): the
    generator's training forward equals its no-grad forward bit for bit (the D-step / G-step pair of pi_GAN/train.py),
    and the backward is exactly linear in the cotangent: the gradients of 2 L are 2 x the gradients of L to the bit
    (every kernel of the chain, the dW GEMMs and the reductions is linear in dL/d(raw) and scaling by 2 is exact in
    fp32), whatever split into kept / recomputed ray ranges the memory planner chose."""
    gen = _generator(res, nc, nf)
    z = torch.randn(batch, 256, device=dev(), generator=torch.Generator(device=dev()).manual_seed(3))
    rng = np.random.Generator(np.random.PCG64(5))                # fixed poses (the Renderer draws them from NumPy's global RNG)
    thetas, phis = list(rng.normal(0, 0.3, batch)), list(rng.normal(0, 0.15, batch))
    with torch.no_grad():
        ref_img = gen(z, thetas, phis, seed=11)
    grads = []
    for scale in (1.0, 2.0):
        for p in gen.parameters():
            p.grad = None
        img = gen(z, thetas, phis, seed=11)
        assert torch.equal(img.detach(), ref_img)
        (scale * torch.nn.functional.softplus(-img.mean(dim=(1, 2, 3))).mean()).backward()
        grads.append([p.grad.clone() for p in gen.film_siren_nerf.parameters()])
    assert tuple(ref_img.shape) == (batch, 3, res, res) and bool(torch.isfinite(ref_img).all())
    for g1, g2 in zip(*grads):
        assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
        assert torch.equal(2.0 * g1, g2)
    torch.cuda.empty_cache()
