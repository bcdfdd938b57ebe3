"""GPU: the library never allocates - callers size every buffer from its *_floats() / *_bytes() queries.  With
MI_DEBUG_GUARDS=1 the binding appends a sentinel zone to the training buffers (saved layer inputs, per-layer gradients,
backward scratch) and checks it after every call: a kernel writing past what a query promised fails here instead of
corrupting a neighbouring tensor.  Ragged point counts on purpose: partial tiles, partial slabs, one-slab passes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import render_ref as R, synth  # noqa: E402


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("kind", ["nerf", "tiny_nerf", "siren_nerf", "film_siren_nerf", "film_siren_nerf_nodir"])
@pytest.mark.parametrize("n,s", [(1, 3), (7, 19), (300, 32), (129, 64), (1024, 24), (37, 192)])
def test_training_buffers_are_never_overrun(monkeypatch, kind, n, s):
    from mirender import autograd as A, fields, ops
    monkeypatch.setenv("MI_DEBUG_GUARDS", "1")
    m = fields.field_from_state_dict(synth.state_dict(kind, seed=3, sharp="medium", bias_jitter=0.05), dev())
    pf = fields.as_packed_field(m)
    rays = torch.from_numpy(R.rays_from_camera(40, 40, 55.0, synth.pose_degrees(4.0, 10.0, -30.0))[:n]).to(dev())
    rng = np.random.Generator(np.random.PCG64(n * 1000 + s))
    z = torch.from_numpy(np.sort(rng.uniform(2, 6, size=(n, s)).astype(np.float32), -1)).to(dev())
    film = synth.film_params(1, seed=2).to(dev()) if kind.startswith("film") else None
    raw, saved = A._forward_pass(pf, rays, z, film, 1 << 40)          # guarded saved-input buffer
    assert torch.equal(raw, ops.field_eval_rays(pf, rays, z, film))
    g_raw = torch.from_numpy(rng.normal(size=(n, s, 4)).astype(np.float32)).to(dev())
    got, got_film = A._field_backward(pf, rays, z, raw, g_raw, film, saved)      # guarded gradient rows + scratch
    got2, _ = A._field_backward(pf, rays, z, raw, g_raw, film)                    # ... and the recompute path
    torch.cuda.synchronize()
    for a, b in zip(got, got2):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


@pytest.mark.parametrize("kind", ["nerf", "film_siren_nerf"])
@pytest.mark.parametrize("n,nc,nf", [(1, 3, 0), (2, 3, 1), (129, 16, 24), (1000, 64, 128), (4097, 12, 24), (63, 256, 256)])
def test_render_workspace_is_never_overrun(monkeypatch, kind, n, nc, nf):
    """mi_render_rays carves depths, raw outputs and weights of both passes out of ONE caller-provided workspace
    (mi_render_workspace_bytes): a sentinel zone behind it must survive the call, and the result must not depend on
    the workspace (guarded fresh buffer vs the cached grow-only one)."""
    from mirender import fields, ops
    m = fields.field_from_state_dict(synth.state_dict(kind, seed=3, sharp="medium", bias_jitter=0.05), dev())
    pf = fields.as_packed_field(m)
    rays = torch.from_numpy(R.rays_from_camera(80, 80, 111.0, synth.pose_degrees(4.0, 10.0, -30.0))[:n]).to(dev())
    film = synth.film_params(1, seed=2).to(dev()) if kind.startswith("film") else None
    near, far = (0.5, 1.5) if film is not None else (2.0, 6.0)
    plain = ops.render_rays_fused(pf, pf, rays, near, far, nc, nf, film, seed=5)
    monkeypatch.setenv("MI_DEBUG_GUARDS", "1")
    guarded = ops.render_rays_fused(pf, pf, rays, near, far, nc, nf, film, seed=5)
    for a, b in zip(plain, guarded):
        assert torch.equal(a, b)


def test_guard_catches_an_overrun(monkeypatch):
    """The check itself: a buffer whose sentinel zone was touched is reported."""
    from mirender import _lib, autograd as A
    monkeypatch.setenv("MI_DEBUG_GUARDS", "1")
    view, whole = A._guarded(100, dev())
    assert whole is not None and view.numel() == 100
    A._check_guard(whole, "untouched")
    whole[100 + 5] = 0.0
    with pytest.raises(_lib.MiRenderError, match="past the end"):
        A._check_guard(whole, "touched")
