"""GPU: windows of the REAL full-size launches of BASELINE configs C3 / C4 / C5 against the oracle (VERDICT r02 "next" #2).

test_gpu_fullsize.py covers the full sizes through properties, test_gpu_render.py checks separately launched 1 024-ray
windows against the oracle.  Here the two meet: ONE full-size launch (640 000 rays at 64+128; the batch-32 generator
at 128x128; a GPU's four 256x256 images at 24+48) with injected jitter, whose stage chain must equal the fused call
bit for bit over the WHOLE ray list, and rows cut out of THAT launch's outputs and intermediates go through
oracle/parity.py:check_render (coarse pass, resampling on the launch's own weights, fine pass at the launch's own
depths: hard 1e-4 gates, the fp64 bound for the x50 heads) - the oracle only ever sees the window's rays.

For the pi_GAN launches the backward is checked the same way: a cotangent that is non-zero on the window only makes
every gradient of the full-size backward (all field weights and the window image's FiLM row) a function of the
window's rays alone, so the oracle's autograd on those 256 rays at the launch's own fine depths is the reference
(nerf/render.py:150-167, pi_GAN/modules.py:176-184, pi_GAN/render.py:195-206).  Exact linearity in the cotangent ties
that to the launch with a dense cotangent: grad(window) + grad(rest) == grad(all) up to the rounding of the sums.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fields as ofields, parity, render_ref as R, synth  # noqa: E402


def dev():
    return torch.device("cuda", 0)


def _window(chain: dict, lo: int, hi: int) -> dict:
    return {k: v[lo:hi] for k, v in chain.items()}


@pytest.mark.parametrize("sharp", ["medium", True])
def test_c3_full_launch_window(sharp):
    """C3: one 800x800 frame, 64+128 samples, separate coarse / fine NeRF 8x256, t_rand [640 000, 64] injected; rows
    320 200 : 321 224 of that launch."""
    from mirender import fields, ops, render_core
    W = H = 800
    nc, nf, near, far = 64, 128, 2.0, 6.0
    lo, hi = 320200, 321224
    sd_c = synth.state_dict("nerf", seed=0, sharp=sharp, bias_jitter=0.05)
    sd_f = synth.state_dict("nerf", seed=1, sharp=sharp, bias_jitter=0.05)
    cm, fm = fields.field_from_state_dict(sd_c, dev()), fields.field_from_state_dict(sd_f, dev())
    pose, focal = synth.pose_degrees(4.0, 0.0, -30.0), 1.3875 * W
    tr = synth.t_rand(W * H, nc, seed=321)
    tr_d = tr.to(dev())
    with torch.no_grad():
        image = render_core._render_image_device(W, H, focal, pose, near, far, cm, fm, nc, nf, None, tr_d, None, 0, W * H)
        rays = ops.gen_rays(W, H, focal, pose, dev())
        fused = render_core.render_rays(rays, near, far, cm, fm, nc, nf, t_rand=tr_d)
        chain = parity.hip_stage_chain(ops, fields.as_packed_field(cm), fields.as_packed_field(fm), rays, near, far, nc, nf, tr_d)
    assert rays.shape[0] == W * H
    parity.assert_chain_equals_fused(chain, fused)                  # over all 640 000 rays of the launch
    for a, b in zip(image, fused[3:6]):
        assert torch.equal(a, b)                                    # the image-level entry point is that same launch
    rays_w = torch.from_numpy(R.rays_from_camera(W, H, focal, pose)[lo:hi])
    assert torch.equal(rays[lo:hi].cpu(), rays_w)
    fc, ff = ofields.make_field("nerf", sd_c), ofields.make_field("nerf", sd_f)
    f64 = (ofields.make_field("nerf", {k: v.double() for k, v in sd_c.items()}),
           ofields.make_field("nerf", {k: v.double() for k, v in sd_f.items()}))
    with torch.no_grad():
        ref = R.render_rays(rays_w, near, far, fc, ff, nc, nf, tr[lo:hi])
    case = f"C3 full launch window: 800x800 64+128 one launch, rows {lo}:{hi}, sharp={sharp}"
    rec = parity.check_render(case, _window(chain, lo, hi), ref, f64, rays_w, near, far, nc, nf, tr[lo:hi], fc, ff,
                              sharp=sharp is True)
    assert rec["psnr_vs_oracle"] >= 54.3, rec
    del chain, fused, image
    ops._Workspace.release()
    torch.cuda.empty_cache()


def _generator(res, nc, nf):
    from mirender import pigan
    torch.manual_seed(0)
    gen = pigan.Generator(256, res, near=0.5, far=1.5, fov=12, coarse_samples=nc, fine_samples=nf).to(dev())
    gen.film_siren_nerf.load_state_dict(synth.state_dict("film_siren_nerf", seed=40, sharp="medium"))
    return gen


@pytest.mark.parametrize("name,res,batch,nc,nf,image,row", [("C4", 128, 32, 12, 24, 17, 60), ("C5 per GPU", 256, 4, 24, 48, 2, 131)])
def test_pigan_full_launch_window(name, res, batch, nc, nf, image, row):
    """C4 (128x128, batch 32, 12+24) / one GPU's share of C5 (256x256, 4 of the global 32 images, 24+48): a 256-ray
    window of one image in the middle of the batch - forward link by link, and every gradient of the full-size
    backward (field weights, that image's FiLM row) for a cotangent living on the window."""
    from mirender import fields, ops, pigan, render_core
    near, far = 0.5, 1.5
    gen = _generator(res, nc, nf)
    net = gen.film_siren_nerf
    pf = fields.as_packed_field(net)
    per = res * res
    n = batch * per
    lo = image * per + row * res + 32 if res == 128 else image * per + row * res
    hi = lo + 256
    rng = np.random.Generator(np.random.PCG64(5))
    thetas, phis = list(rng.normal(0, 0.3, batch)), list(rng.normal(0, 0.15, batch))
    z = torch.randn(batch, 256, device=dev(), generator=torch.Generator(device=dev()).manual_seed(3))
    tr = synth.t_rand(n, nc, seed=77)
    tr_d = tr.to(dev())
    film = gen.get_mapping(z).detach().requires_grad_(True)                  # [b, 9, 512]: leaf, so its rows' gradients show
    focal = gen.renderer.focal
    poses = [pigan.camera_pos_to_transform_matrix(1, thetas[i], phis[i]) for i in range(batch)]

    def forward():
        return pigan.render_batch(net, film, poses, res, res, focal, near, far, nc, nf, t_rand=tr_d)     # [b, H, W, 3], graph on

    img = forward()
    with torch.no_grad():
        rays = torch.cat([ops.gen_rays(res, res, focal, p, dev()) for p in poses])
        fused = render_core.render_rays(rays, near, far, net, net, nc, nf, t_rand=tr_d, film=film.detach())
        chain = parity.hip_stage_chain(ops, pf, pf, rays, near, far, nc, nf, tr_d, film.detach())
    parity.assert_chain_equals_fused(chain, fused)                          # the whole batch, every ray
    assert torch.equal(img.detach().reshape(-1, 3), fused[3])               # training forward == inference forward

    # ---- forward: the window's rows of that launch against the oracle ------------------------------------------------
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    film_w = film.detach()[image].cpu()
    rays_w = torch.from_numpy(R.rays_from_camera(res, res, focal, poses[image]))[lo - image * per:hi - image * per]
    assert torch.equal(rays[lo:hi].cpu(), rays_w)
    fo = ofields.make_field("film_siren_nerf", sd, film_w)
    with torch.no_grad():
        ref = R.render_rays(rays_w, near, far, fo, fo, nc, nf, tr[lo:hi])
    case = f"{name} full launch window: {res}x{res} batch {batch} {nc}+{nf} one launch, image {image} rays {lo - image * per}:{hi - image * per}"
    rec = parity.check_render(case, _window(chain, lo, hi), ref, None, rays_w, near, far, nc, nf, tr[lo:hi], fo, fo)
    assert rec["psnr_vs_oracle"] >= 54.3, rec

    # ---- backward: cotangent on the window only -> every gradient of the full-size backward depends on those rays alone --
    cot_w = torch.from_numpy(rng.normal(size=(256, 3)).astype(np.float32))
    cot_full = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)).to(dev())
    cot_win = torch.zeros_like(cot_full)
    cot_win[lo:hi] = cot_w.to(dev())
    cot_rest = cot_full.clone()
    cot_rest[lo:hi] = 0
    cot_all = cot_rest + cot_win
    grads = {}
    params = list(net.parameters())
    for tag, cot in (("window", cot_win), ("rest", cot_rest), ("all", cot_all)):
        for p in params:
            p.grad = None
        film.grad = None
        out = forward()
        (out.reshape(-1, 3) * cot).sum().backward()
        grads[tag] = [p.grad.clone() for p in params] + [film.grad.clone()]
    g_film = grads["window"][-1]
    others = torch.ones(batch, dtype=torch.bool)
    others[image] = False
    assert float(g_film[others.to(dev())].abs().max()) == 0.0               # no other image's FiLM row hears the window
    # oracle autograd, fp32 and fp64, on the 256 rays at THE LAUNCH'S OWN fine depths (the resampling carries no gradient:
    # render.py:141 detaches z_samples)
    z_w = chain["z_fine"][lo:hi].cpu()
    refs = {}
    for dt in (torch.float32, torch.float64):
        sd_req = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
        film_req = film_w.clone().to(dt).requires_grad_(True)
        ro, rd = rays_w[:, 0].to(dt), rays_w[:, 1].to(dt)
        raw = R.query_field(R.points_on_rays(ro, rd, z_w.to(dt)), rd / torch.norm(rd, dim=-1, keepdim=True),
                            ofields.make_field("film_siren_nerf", sd_req, film_req))
        rgb, _, _, _ = R.composite(raw, z_w.to(dt), rd)
        (rgb * cot_w.to(dt)).sum().backward()
        refs[dt] = {k: v.grad for k, v in sd_req.items()}
        refs[dt]["film row"] = film_req.grad
    names = [k for k, _ in net.named_parameters()]
    got = dict(zip(names, grads["window"][:-1]))
    got["film row"] = g_film[image]
    recs = [parity.gate_grad(case, k, got[k].cpu(), refs[torch.float32][k], refs[torch.float64][k],
                             tol=parity.GRAD_TOL_SMOOTH, elem_tol=parity.GRAD_ELEM_TOL_SMOOTH, check=False,
                             stage="gradient (full-size backward, cotangent on the window)") for k in got]
    bad = [r for r in recs if not r["passed"]]
    assert not bad, bad[0]
    # linearity: the launch with a dense cotangent is the sum of the window's and the rest's (fp32 sums over ~10^7 points
    # in different groupings: agreement to a few 1e-5 of each tensor's largest entry, not bit equality)
    worst = 0.0
    for a, b, c in zip(grads["window"], grads["rest"], grads["all"]):
        worst = max(worst, float((a + b - c).abs().max()) / max(float(c.abs().max()), 1e-30))
    parity.record(case=case, stage="gradient (linearity in the cotangent)", qty="grad(window) + grad(rest) - grad(all)",
                  err_vs_oracle32=worst, tol=1e-4, unit="max abs / max |grad(all)| per tensor, worst tensor", active="hard",
                  passed=worst <= 1e-4)
    assert worst <= 1e-4, worst
    torch.cuda.empty_cache()
