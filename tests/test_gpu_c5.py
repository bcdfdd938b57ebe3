"""GPU: BASELINE config C5 - the pi_GAN 256x256 training step at "48 samples/ray" (Nc=24, Nf=48),
data-parallel with a gradient all-reduce - at sizes the oracle's autograd finishes in seconds:

  * the FiLM training path at 24+48 samples with the memory planner forced to split ONE image into several ray
    ranges (what 256x256 at 72 samples needs: mirender/autograd.py:_chunk_ranges), some kept, some recomputed:
    gradients against the oracle's autograd (fp64 as the truth, the fp32 oracle's own distance as the scale);
  * one whole training step as pi_GAN/train.py:99-136 issues the generator - D-step forward without a graph,
    G-step forward + backward - through Generator, split and unsplit;
  * the same G-step on two ranks (one device, gloo transport) with images sharded over ranks and ONE flat
    gradient all-reduce: averaged gradients equal the full-batch ones.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import fields as ofields, parity, render_ref as R, synth  # noqa: E402

NC, NF = 24, 48
RES = 12                      # 144 rays per image, 10 368 fine-pass points


def dev():
    return torch.device("cuda", 0)


def _film_model(sharp="medium"):
    from mirender import fields
    m = fields.FilmSirenNeRF().to(dev())
    m.load_state_dict(synth.state_dict("film_siren_nerf", seed=40, sharp=sharp))
    return m


def test_c5_film_grads_with_an_image_split_into_ranges(monkeypatch):
    from mirender import autograd as A, fields, ops, pigan
    kind, b = "film_siren_nerf", 2
    n = b * RES * RES
    sd = synth.state_dict(kind, seed=40, sharp="medium")
    m = _film_model()
    pf = fields.as_packed_field(m)
    focal = RES / 2 / np.tan(12 / 2 * np.pi / 180)
    poses = [pigan.camera_pos_to_transform_matrix(1, t, p) for t, p in ((0.2, -0.1), (-0.15, 0.05))]
    rays = torch.cat([torch.from_numpy(R.rays_from_camera(RES, RES, focal, p)) for p in poses])
    rng = np.random.Generator(np.random.PCG64(1))
    z = torch.from_numpy(np.sort(rng.uniform(0.5, 1.5, size=(n, NC + NF)).astype(np.float32), -1))
    cot = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32))         # pi_GAN consumes rgb_fine only
    film0 = synth.film_params(b, seed=8)
    refs = {}
    for dt in (torch.float32, torch.float64):
        sd_req = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
        film_req = film0.clone().to(dt).requires_grad_(True)
        ro, rd = rays[:, 0].to(dt), rays[:, 1].to(dt)
        pts, view = R.points_on_rays(ro, rd, z.to(dt)), rd / torch.norm(rd, dim=-1, keepdim=True)
        per = RES * RES
        raw = torch.cat([R.query_field(pts[i * per:(i + 1) * per], view[i * per:(i + 1) * per],
                                       ofields.make_field(kind, sd_req, film_req[i])) for i in range(b)])
        rgb, _, _, _ = R.composite(raw, z.to(dt), rd)
        (rgb * cot.to(dt)).sum().backward()
        refs[dt] = {k: v.grad.double() for k, v in sd_req.items()}
        refs[dt]["__film__"] = film_req.grad.double()

    rays_d, z_d, film_d = rays.to(dev()), z.to(dev()), film0.to(dev())
    names = []
    for key, _ in fields.SPECS[pf.kind]:
        names += [key + ".weight", key + ".bias"]
    results = {}
    # one image per range | 4 parts per image, nothing kept | 4 parts per image, 3 of the 8 parts kept
    for tag, max_rays, keep_parts in (("whole images", 144, 0), ("split, recomputed", 40, 0), ("split, 3 kept", 40, 3)):
        monkeypatch.setattr(A, "_max_points_per_chunk", lambda pf_, mr=max_rays: mr * (NC + NF))
        per_point = 4 * A._lib.load().mi_field_train_acts_floats(pf.kind)
        monkeypatch.setattr(A, "SAVE_FINE_BYTES", per_point * 36 * (NC + NF) * keep_parts)
        _, _, ranges = A._chunk_ranges(pf, n, NC + NF, film_d)
        assert len(ranges) == (2 if max_rays == 144 else 8) and ranges[-1][1] == n
        raw_d, saved = A._forward_pass(pf, rays_d, z_d, film_d, A.SAVE_FINE_BYTES)
        assert len(saved) == keep_parts
        plain = ops.field_eval_rays(pf, rays_d, z_d, film_d)
        assert torch.equal(raw_d, plain)                       # saving / plain / mixed forwards agree bit for bit
        g_raw = A._composite_bwd(raw_d, z_d, rays_d, cot.to(dev()), None, None)
        got, got_film = A._field_backward(pf, rays_d, z_d, raw_d, g_raw, film_d, saved)
        results[tag] = list(got) + [got_film]
        recs = [parity.gate_grad(f"C5 film grads 24+48 [{tag}]", name, t.cpu(), refs[torch.float32][name],
                                 refs[torch.float64][name], tol=parity.GRAD_TOL_SMOOTH,
                                 elem_tol=parity.GRAD_ELEM_TOL_SMOOTH, check=False, stage="gradient (field backward)")
                for name, t in list(zip(names, got)) + [("__film__", got_film)]]
        bad = [r for r in recs if not r["passed"]]
        assert not bad, bad[0]
    for a_, b_ in zip(results["split, recomputed"], results["split, 3 kept"]):
        assert torch.equal(a_, b_)                             # kept or recomputed: the same ranges, the same sums
    for a_, b_ in zip(results["whole images"], results["split, recomputed"]):
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(1e-3, float(a_.abs().max()))


def _generator():
    from mirender import pigan
    torch.manual_seed(0)
    gen = pigan.Generator(32, RES, near=0.5, far=1.5, fov=12, coarse_samples=NC, fine_samples=NF)
    gen.film_siren_nerf.load_state_dict(synth.state_dict("film_siren_nerf", seed=40, sharp="medium"))
    rng = np.random.Generator(np.random.PCG64(2))
    with torch.no_grad():
        for p in gen.mapping_network.parameters():
            p.copy_(torch.from_numpy(rng.uniform(-0.08, 0.08, size=tuple(p.shape)).astype(np.float32)))
        for head in gen.mapping_network.output_layers:
            head.bias[:256] += 1.0
    return gen.to(dev())


B = 4
THETAS, PHIS = [0.2, -0.1, 0.05, 0.3], [0.0, 0.1, -0.05, -0.12]


def _inputs():
    z = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).standard_normal((B, 32)).astype(np.float32)).to(dev())
    tr = synth.t_rand(B * RES * RES, NC, seed=11).to(dev())
    return z, tr


def _g_step(gen, z, tr, lo, hi):
    """The generator's part of one training step (pi_GAN/train.py:99-136) on images [lo, hi): D-step forward with
    requires_grad off (no graph), G-step forward + backward of a mean-over-images loss."""
    from mirender import dist as mdist  # noqa: F401
    per = RES * RES
    args = (z[lo:hi], THETAS[lo:hi], PHIS[lo:hi])
    for p in gen.parameters():
        p.requires_grad_(False)                                  # utils.requires_grad(generator, False), train.py:101
    d_img = gen(*args, t_rand=tr[lo * per:hi * per])
    assert not d_img.requires_grad
    for p in gen.parameters():
        p.requires_grad_(True)
        p.grad = None
    g_img = gen(*args, t_rand=tr[lo * per:hi * per])
    assert g_img.requires_grad and torch.equal(g_img.detach(), d_img)    # the two forwards are the same function
    loss = torch.nn.functional.softplus(-(g_img * 3.0).mean(dim=(1, 2, 3))).mean()
    loss.backward()
    return g_img.detach(), loss.detach()


def test_c5_training_step_through_generator_split_and_unsplit(monkeypatch):
    from mirender import autograd as A
    z, tr = _inputs()
    gen = _generator()
    img0, loss0 = _g_step(gen, z, tr, 0, B)
    g0 = [p.grad.clone() for p in gen.parameters()]
    assert tuple(img0.shape) == (B, 3, RES, RES) and all(torch.isfinite(g).all() for g in g0)
    assert float(gen.mapping_network.output_layers[4].weight.grad.abs().max()) > 0
    monkeypatch.setattr(A, "_max_points_per_chunk", lambda pf_: 40 * (NC + NF))       # 4 parts per image
    per_point = 4 * A._lib.load().mi_field_train_acts_floats(2)
    monkeypatch.setattr(A, "SAVE_FINE_BYTES", per_point * 36 * (NC + NF) * 5)          # 5 of the 16 parts kept
    img1, loss1 = _g_step(gen, z, tr, 0, B)
    assert torch.equal(img0, img1) and torch.equal(loss0, loss1)
    for a_, p in zip(g0, gen.parameters()):
        assert float((a_ - p.grad).abs().max()) <= 2e-5 * max(1e-3, float(a_.abs().max()))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirender import dist as mdist
    z, tr = _inputs()
    gen = _generator()
    lo, hi = mdist.shard_range(B, rank, world)                    # images sharded along dim 0 like DataParallel
    img, loss = _g_step(gen, z, tr, lo, hi)
    mdist.allreduce_grads(list(gen.parameters()))
    out[rank] = (img.cpu().numpy(), [p.grad.cpu().numpy() for p in gen.parameters()])
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_c5_two_rank_data_parallel_generator_step():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    z, tr = _inputs()
    gen = _generator()
    img, _ = _g_step(gen, z, tr, 0, B)
    full = [p.grad.cpu().numpy() for p in gen.parameters()]
    imgs = np.concatenate([out[r][0] for r in range(world)])
    assert np.array_equal(imgs, img.cpu().numpy())                 # each rank rendered exactly its images
    for r in range(world):
        for g, ref in zip(out[r][1], full):
            assert np.abs(g - ref).max() <= 2e-5 * max(1e-3, np.abs(ref).max())
