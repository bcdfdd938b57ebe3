"""GPU: ONE field for both passes (pi_GAN: `render_image(..., model, model, ...)`, pi_GAN/modules.py:160-161; nerf with
use_fine_model off, nerf/train_nerf.py:91,94).

The fine pass of render_rays (nerf/render.py:143-144) evaluates the field at sort(cat(z_coarse, z_samples)); Nc of those
Nc + Nf points are the points the coarse pass already evaluated with the SAME field.  The renderer evaluates the Nf new
depths only and merges (`mi_sample_fine_pos`, `mi_merge_raw`, in backward `mi_split_grad`).  That must change nothing:

* inference: the fused call (shared path) equals the stage chain that evaluates all Nc + Nf points, bit for bit, for
  every field kind, ragged ray counts, Nf = 0 (the alias of SURVEY.md 8d C2) and with a workspace too small for the
  shared path (the library then takes the plain path);
* training: outputs bit-equal and gradients equal (up to the order of two partial sums) to the same call made with
  two PackedField views of the same parameters, which takes the two-field path and evaluates everything twice;
* the position table is a permutation that sorts: z_fine[pos[e]] is the e-th input depth."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import parity, render_ref as R, synth  # noqa: E402


def dev():
    return torch.device("cuda", 0)


def _field(kind, seed=5):
    from mirender import fields
    return fields.field_from_state_dict(synth.state_dict(kind, seed=seed, sharp="medium", bias_jitter=0.05), dev())


def _rays(n, film):
    pose = synth.pose_radians(1.0, 0.2, -0.1) if film else synth.pose_degrees(4.0, 20.0, -30.0)
    return torch.from_numpy(R.rays_from_camera(40, 40, 180.0 if film else 55.0, pose)[:n]).to(dev())


@pytest.mark.parametrize("kind,n,nc,nf", [("film_siren_nerf", 256, 12, 24), ("film_siren_nerf", 130, 24, 48),
                                          ("film_siren_nerf_nodir", 64, 6, 6), ("nerf", 257, 64, 128), ("siren_nerf", 33, 8, 1),
                                          ("tiny_nerf", 500, 32, 0), ("film_siren_nerf", 128, 12, 0)])
def test_shared_field_inference_equals_evaluating_every_point(kind, n, nc, nf):
    from mirender import _lib, fields, ops
    is_film = kind.startswith("film")
    m = _field(kind)
    pf = fields.as_packed_field(m)
    film = synth.film_params(2 if is_film and n % 2 == 0 else 1, seed=3).to(dev()) if is_film else None
    near, far = (0.5, 1.5) if is_film else (2.0, 6.0)
    rays, tr = _rays(n, is_film), synth.t_rand(n, nc, seed=9).to(dev())
    with torch.no_grad():
        fused = ops.render_rays_fused(pf, pf, rays, near, far, nc, nf, film, tr)                 # shared path
        chain = parity.hip_stage_chain(ops, pf, pf, rays, near, far, nc, nf, tr, film)            # every point, twice
    parity.assert_chain_equals_fused(chain, fused)
    # a second PackedField over the same parameters is a different object: the two-field path of the fused call
    pf2 = fields.PackedField(pf.kind, pf.params)
    with torch.no_grad():
        plain = ops.render_rays_fused(pf, pf2, rays, near, far, nc, nf, film, tr)
    for a, b in zip(fused, plain):
        assert torch.equal(a, b)
    # the C ABI with a workspace that only holds the base regions: same field, plain path, same results
    lib = _lib.load()
    base = lib.mi_render_workspace_bytes(n, nc, nf)
    ws = torch.empty(base, dtype=torch.uint8, device=dev())
    outs = [torch.empty(s, dtype=torch.float32, device=dev()) for s in ((n, 3), (n,), (n,), (n, 3), (n,), (n,))]
    groups = 1 if film is None else film.shape[0]
    zl, ul = ops.linspace_table(near, far, nc, dev()), ops.linspace_table(0.0, 1.0, nf, dev())
    rc = lib.mi_render_rays(pf.kind, _lib.ptr(pf.refresh()), pf.kind, _lib.ptr(pf.refresh()), _lib.ptr(film), _lib.ptr(rays),
                            groups, n // groups, near, far, nc, nf, _lib.ptr(zl), _lib.ptr(ul), _lib.ptr(tr), 0, 0,
                            *[_lib.ptr(o) for o in outs], _lib.ptr(ws), base, _lib.stream_ptr(dev()))
    assert rc == 0, lib.mi_last_error()
    for a, b in zip(fused, outs):
        assert torch.equal(a, b)
    rc = lib.mi_render_rays(pf.kind, _lib.ptr(pf.refresh()), pf.kind, _lib.ptr(pf.refresh()), _lib.ptr(film), _lib.ptr(rays),
                            groups, n // groups, near, far, nc, nf, _lib.ptr(zl), _lib.ptr(ul), _lib.ptr(tr), 0, 0,
                            *[_lib.ptr(o) for o in outs], _lib.ptr(ws), base - 1, _lib.stream_ptr(dev()))
    assert rc == -1 and b"workspace" in lib.mi_last_error()           # smaller than the base regions: refused


def test_positions_are_the_sorting_permutation_and_merge_split_are_transposes():
    from mirender import ops
    n, nc, nf = 301, 16, 40
    rng = np.random.Generator(np.random.PCG64(4))
    z_c = torch.from_numpy(np.sort(rng.uniform(2, 6, (n, nc)).astype(np.float32), -1)).to(dev())
    w = torch.from_numpy(rng.random((n, nc), dtype=np.float32) ** 4).to(dev())
    w[7] = 0                                                               # all-zero weights: uniform resampling
    z_c[11, 5:9] = z_c[11, 5]                                              # repeated depths
    z_f, z_s, pos = ops.sample_fine_pos(z_c, w, 2.0, 6.0, nf)
    assert torch.equal(z_f, ops.sample_fine(z_c, w, 2.0, 6.0, nf))
    zall = torch.cat([z_c, z_s], 1)
    p = pos.long()
    assert torch.equal(torch.sort(p, 1).values, torch.arange(nc + nf, device=dev()).expand(n, -1))      # a permutation
    assert torch.equal(torch.gather(z_f, 1, p), zall)                                                   # that sorts
    raw_c, raw_s = torch.randn(n, nc, 4, device=dev()), torch.randn(n, nf, 4, device=dev())
    raw_f = ops.merge_raw(raw_c, raw_s, pos)
    want = torch.empty(n, nc + nf, 4, device=dev())
    want.scatter_(1, p[:, :, None].expand(-1, -1, 4), torch.cat([raw_c, raw_s], 1))
    assert torch.equal(raw_f, want)
    g_f = torch.randn(n, nc + nf, 4, device=dev())
    g_c, g_s = ops.split_grad(g_f, pos, nc)
    both = torch.gather(g_f, 1, p[:, :, None].expand(-1, -1, 4))
    assert torch.equal(g_c, both[:, :nc]) and torch.equal(g_s, both[:, nc:])
    base = torch.randn(n, nc, 4, device=dev())
    acc, _ = ops.split_grad(g_f, pos, nc, base.clone())
    assert torch.equal(acc, base + both[:, :nc])


@pytest.mark.parametrize("kind,nc,nf,coarse_loss", [("film_siren_nerf", 12, 24, False), ("film_siren_nerf", 12, 24, True),
                                                   ("tiny_nerf", 16, 16, True), ("siren_nerf", 8, 12, False),
                                                   ("tiny_nerf", 16, 0, True)])
def test_shared_field_training_equals_the_two_field_path(kind, nc, nf, coarse_loss):
    """The same parameters seen as ONE PackedField (shared path: Nf new depths + merge, gradients split back) and as
    TWO (every point evaluated twice, as the reference does): outputs bit-equal, gradients equal up to the order in
    which two partial sums are added (5e-6 of each tensor's largest entry)."""
    from mirender import autograd as A, fields
    is_film = kind.startswith("film")
    m = _field(kind, seed=8)
    pf = fields.as_packed_field(m)
    pf2 = fields.PackedField(pf.kind, pf.params)
    n = 192
    film = synth.film_params(2, seed=4).to(dev()).requires_grad_(True) if is_film else None
    near, far = (0.5, 1.5) if is_film else (2.0, 6.0)
    rays, tr = _rays(n, is_film), synth.t_rand(n, nc, seed=2).to(dev())
    cot = [torch.randn(s, device=dev(), generator=torch.Generator(device=dev()).manual_seed(i)) for i, s in
           enumerate(((n, 3), (n,), (n,), (n, 3), (n,), (n,)))]
    results = []
    for fine_view in (pf, pf2):
        for p in m.parameters():
            p.grad = None
        if film is not None:
            film.grad = None
        out = A.render_rays_train(pf, fine_view, rays, near, far, nc, nf, film, tr, 0)
        loss = sum((o * c).sum() for o, c in zip(out[3:], cot[3:]))
        if coarse_loss:
            loss = loss + sum((o * c).sum() for o, c in zip(out[:3], cot[:3]))
        loss.backward()
        if fine_view is pf2:           # two views of the same parameters: autograd already summed their gradients
            pass
        results.append(([o.detach().clone() for o in out], [p.grad.clone() for p in m.parameters()],
                        None if film is None else film.grad.clone()))
    (o1, g1, f1), (o2, g2, f2) = results
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    names = [k for k, _ in m.named_parameters()]
    worst = 0.0
    for name, a, b in zip(names, g1, g2):
        worst = max(worst, float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30))
    if f1 is not None:
        worst = max(worst, float((f1 - f2).abs().max()) / max(float(f2.abs().max()), 1e-30))
    parity.record(case=f"shared field vs two-field path, {kind} {n} rays {nc}+{nf} coarse_loss={coarse_loss}",
                  stage="gradient (shared-field path)", qty="all parameter gradients (+ FiLM table)", err_vs_oracle32=worst,
                  tol=5e-6, unit="max abs / max |grad| per tensor, worst tensor", reference="the two-field HIP path", active="hard",
                  passed=worst <= 5e-6)
    assert worst <= 5e-6, worst
