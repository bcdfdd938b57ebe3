"""Differentiable render_rays: what PyTorch autograd provides in the reference (nerf/train_nerf.py:151-168,
pi_GAN/modules.py:159-161), rebuilt on the HIP kernels.

Forward = the inference kernels (nothing but the small per-sample tensors z / raw is kept).  Backward, per
pass that received a gradient: compositing backward -> dL/d(raw); then, in chunks of rays so memory stays
bounded whatever the batch, the field forward is re-run saving each layer's input, the backward chain
produces every layer's dA, and the point-contraction GEMMs reduce them to the weight gradients
(csrc/field_mlp_bwd.hip).  z_samples is detached like the reference (render.py:141), so the fine depths
carry no gradient into the coarse pass.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib, fields, ops

CHUNK_BYTES = 48 << 30             # saved activations + per-layer gradients per backward chunk (288 GB of HBM)
SAVE_IN_FORWARD_BYTES = 48 << 30   # keep layer inputs from the forward when they fit (no recompute in backward)


def _max_points_per_chunk(pf) -> int:
    lib = _lib.load()
    per_point = 4 * (lib.mi_field_train_acts_floats(pf.kind) + lib.mi_field_train_grads_floats(pf.kind))
    return max(4096, CHUNK_BYTES // per_point)


def _groups(pf, film, n_rays):
    if not fields.is_film(pf.kind):
        return None, 1, n_rays
    f = film.detach().to(device=pf.device, dtype=torch.float32).contiguous().reshape(-1, 9, 512)
    if n_rays % f.shape[0]:
        raise _lib.MiRenderError("rays must split evenly over the FiLM groups (images)")
    return f, f.shape[0], n_rays // f.shape[0]


def _forward_saving(pf: fields.PackedField, rays, z, film):
    """Field forward that also keeps every layer's input (training forward).  Returns (raw, acts)."""
    lib = _lib.load()
    dev = pf.device
    n, s = z.shape
    pts = n * s
    f, groups, rpg = _groups(pf, film, n)
    acts = torch.empty(lib.mi_field_train_acts_floats(pf.kind) * pts, dtype=torch.float32, device=dev)
    raw = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_field_eval_rays_train(pf.kind, _lib.ptr(pf.refresh()), _lib.ptr(f), _lib.ptr(rays), _lib.ptr(z),
                                                groups, rpg, s, _lib.ptr(raw), _lib.ptr(acts), _lib.stream_ptr(dev)),
                   "mi_field_eval_rays_train")
    return raw, acts


def _can_save(pf: fields.PackedField, n_points: int) -> bool:
    lib = _lib.load()
    a = lib.mi_field_train_acts_floats(pf.kind)
    return a > 0 and 4 * a * n_points <= SAVE_IN_FORWARD_BYTES


def _field_backward(pf: fields.PackedField, rays, z, raw, g_raw, film, acts=None):
    """Gradients of one pass: (list of tensors shaped like pf.params, grad of the FiLM table or None).
    `acts` = layer inputs kept by the forward; without them the forward is re-run chunk by chunk (whole
    FiLM groups per chunk) so memory stays bounded."""
    lib = _lib.load()
    dev = pf.device
    acts_f, grads_f = lib.mi_field_train_acts_floats(pf.kind), lib.mi_field_train_grads_floats(pf.kind)
    n, s = z.shape
    f_all, groups, rpg = _groups(pf, film, n)
    if acts is not None:
        groups_per_chunk = groups
    else:
        groups_per_chunk = max(1, _max_points_per_chunk(pf) // (rpg * s)) if f_all is not None else 1
    rays_per_chunk = groups_per_chunk * rpg if f_all is not None else (n if acts is not None else
                                                                        max(1, _max_points_per_chunk(pf) // s))
    total = None
    g_film = None if f_all is None else torch.empty_like(f_all)
    packed_bwd = pf.refresh_bwd()
    stream = _lib.stream_ptr(dev)
    for r0 in range(0, n, rays_per_chunk):
        r1 = min(n, r0 + rays_per_chunk)
        pts = (r1 - r0) * s
        f_c = g_c = fp = None
        ng, ppg = 1, pts
        if f_all is not None:
            ng = (r1 - r0) // rpg
            ppg = rpg * s
            f_c, g_c = f_all[r0 // rpg:r0 // rpg + ng], g_film[r0 // rpg:r0 // rpg + ng]
            fp = torch.empty(lib.mi_field_film_partial_floats(ng, ppg), dtype=torch.float32, device=dev)
        if acts is not None:
            acts_c, raw_c = acts, raw
        else:
            raw_c, acts_c = _forward_saving(pf, rays[r0:r1], z[r0:r1], f_c)
        gws = torch.empty(grads_f * pts, dtype=torch.float32, device=dev)
        part = torch.empty(lib.mi_field_bwd_partial_floats(pts), dtype=torch.float32, device=dev)
        out = [torch.empty_like(p) for p in pf.params]
        arr = (ctypes.c_void_p * len(out))(*[t.data_ptr() for t in out])
        # FiLM kinds: d gamma = <W, dW_image> + b . db_image needs the parameters themselves
        par = (ctypes.c_void_p * len(out))(*[p.data_ptr() for p in pf.params]) if f_all is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.mi_field_backward(pf.kind, _lib.ptr(packed_bwd), _lib.ptr(f_c), _lib.ptr(acts_c), _lib.ptr(gws),
                                             _lib.ptr(raw_c), _lib.ptr(g_raw[r0:r1]), ng, ppg, _lib.ptr(part),
                                             _lib.ptr(fp), arr, par, len(out), _lib.ptr(g_c), stream), "mi_field_backward")
        if total is None:
            total = out
        else:
            torch._foreach_add_(total, out)
    return total, g_film


def _composite_bwd(raw, z, rays, g_rgb, g_depth, g_acc):
    lib = _lib.load()
    dev = raw.device
    n, s = z.shape
    g_raw = torch.empty_like(raw)
    c = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    g_rgb, g_depth, g_acc = c(g_rgb), c(g_depth), c(g_acc)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_composite_bwd(n, s, _lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays), _lib.ptr(g_rgb),
                                        _lib.ptr(g_depth), _lib.ptr(g_acc), _lib.ptr(g_raw), _lib.stream_ptr(dev)),
                   "mi_composite_bwd")
    return g_raw


class _RenderRaysFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pf_c, pf_f, rays, near, far, nc, nf, film, t_rand, seed, *params):
        dev = pf_c.device
        n = rays.shape[0]
        z_c = ops.sample_coarse(n, near, far, nc, dev, t_rand, seed)
        ctx.acts_c = ctx.acts_f = None
        if _can_save(pf_c, n * nc):
            raw_c, ctx.acts_c = _forward_saving(pf_c, rays, z_c, film)
        else:
            raw_c = ops.field_eval_rays(pf_c, rays, z_c, film)
        rgb_c, depth_c, acc_c, w_c = ops.composite(raw_c, z_c, rays)
        z_f = ops.sample_fine(z_c, w_c, near, far, nf)
        if _can_save(pf_f, n * (nc + nf)):
            raw_f, ctx.acts_f = _forward_saving(pf_f, rays, z_f, film)
        else:
            raw_f = ops.field_eval_rays(pf_f, rays, z_f, film)
        rgb_f, depth_f, acc_f, _ = ops.composite(raw_f, z_f, rays, want_weights=False)
        ctx.pf_c, ctx.pf_f, ctx.film = pf_c, pf_f, None if film is None else film.detach()
        ctx.n_c = len(pf_c.params)
        ctx.save_for_backward(rays, z_c, raw_c, z_f, raw_f)
        ctx.set_materialize_grads(False)
        return rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f

    @staticmethod
    def backward(ctx, g_rgb_c, g_depth_c, g_acc_c, g_rgb_f, g_depth_f, g_acc_f):
        rays, z_c, raw_c, z_f, raw_f = ctx.saved_tensors
        pf_c, pf_f = ctx.pf_c, ctx.pf_f
        same = pf_c is pf_f
        grads_c = grads_f = gfilm_c = gfilm_f = None
        if any(g is not None for g in (g_rgb_c, g_depth_c, g_acc_c)):
            g_raw = _composite_bwd(raw_c, z_c, rays, g_rgb_c, g_depth_c, g_acc_c)
            grads_c, gfilm_c = _field_backward(pf_c, rays, z_c, raw_c, g_raw, ctx.film, ctx.acts_c)
        if any(g is not None for g in (g_rgb_f, g_depth_f, g_acc_f)):
            g_raw = _composite_bwd(raw_f, z_f, rays, g_rgb_f, g_depth_f, g_acc_f)
            grads_f, gfilm_f = _field_backward(pf_f, rays, z_f, raw_f, g_raw, ctx.film, ctx.acts_f)
        ctx.acts_c = ctx.acts_f = None          # release the saved activations
        if same:
            if grads_c is not None and grads_f is not None:
                torch._foreach_add_(grads_c, grads_f)
            merged = grads_c if grads_c is not None else grads_f
            param_grads = tuple(merged) if merged is not None else (None,) * ctx.n_c
        else:
            gc = tuple(grads_c) if grads_c is not None else (None,) * len(pf_c.params)
            gf = tuple(grads_f) if grads_f is not None else (None,) * len(pf_f.params)
            param_grads = gc + gf
        g_film = None
        if ctx.film is not None and ctx.needs_input_grad[7]:
            parts = [g for g in (gfilm_c, gfilm_f) if g is not None]
            if parts:
                g_film = (parts[0] if len(parts) == 1 else parts[0] + parts[1]).reshape(ctx.film.shape)
        return (None,) * 7 + (g_film, None, None) + param_grads


def render_rays_train(pf_c, pf_f, rays, near, far, nc, nf, film, t_rand, seed):
    """render_rays with gradients to the field parameters and the FiLM table (6-tuple like render.py:147)."""
    params = list(pf_c.params) if pf_c is pf_f else list(pf_c.params) + list(pf_f.params)
    return _RenderRaysFn.apply(pf_c, pf_f, rays.detach(), float(near), float(far), int(nc), int(nf), film,
                               None if t_rand is None else t_rand.detach(), int(seed), *params)
