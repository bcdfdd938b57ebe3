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

MAX_POINTS_PER_CHUNK = 1 << 19     # ~10 GB of saved activations + per-layer gradients for a NeRF field


def _field_backward(pf: fields.PackedField, rays, z, raw, g_raw, film):
    """Parameter gradients of one pass.  Returns a list of tensors shaped like pf.params."""
    lib = _lib.load()
    dev = pf.device
    acts_f, grads_f = lib.mi_field_train_acts_floats(pf.kind), lib.mi_field_train_grads_floats(pf.kind)
    if acts_f < 0:
        raise _lib.MiRenderError(f"training through field kind {fields.KIND_NAMES[pf.kind]} is not implemented yet")
    n, s = z.shape
    rays_per_chunk = max(1, MAX_POINTS_PER_CHUNK // s)
    total = [torch.zeros_like(p) for p in pf.params]
    packed, packed_bwd = pf.refresh(), pf.refresh_bwd()
    stream = _lib.stream_ptr(dev)
    for r0 in range(0, n, rays_per_chunk):
        r1 = min(n, r0 + rays_per_chunk)
        pts = (r1 - r0) * s
        acts = torch.empty(acts_f * pts, dtype=torch.float32, device=dev)
        gws = torch.empty(grads_f * pts, dtype=torch.float32, device=dev)
        part = torch.empty(lib.mi_field_bwd_partial_floats(pts), dtype=torch.float32, device=dev)
        raw_chunk = torch.empty((r1 - r0, s, 4), dtype=torch.float32, device=dev)
        out = [torch.empty_like(p) for p in pf.params]
        arr = (ctypes.c_void_p * len(out))(*[t.data_ptr() for t in out])
        with torch.cuda.device(dev):
            _lib.check(lib.mi_field_eval_rays_train(pf.kind, _lib.ptr(packed), None, _lib.ptr(rays[r0:r1]),
                                                    _lib.ptr(z[r0:r1]), 1, r1 - r0, s, _lib.ptr(raw_chunk),
                                                    _lib.ptr(acts), stream), "mi_field_eval_rays_train")
            _lib.check(lib.mi_field_backward(pf.kind, _lib.ptr(packed_bwd), _lib.ptr(acts), _lib.ptr(gws),
                                             _lib.ptr(raw_chunk), _lib.ptr(g_raw[r0:r1]), pts, _lib.ptr(part), arr,
                                             len(out), stream), "mi_field_backward")
        torch._foreach_add_(total, out)
    return total


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
        raw_c = ops.field_eval_rays(pf_c, rays, z_c, film)
        rgb_c, depth_c, acc_c, w_c = ops.composite(raw_c, z_c, rays)
        z_f = ops.sample_fine(z_c, w_c, near, far, nf)
        raw_f = ops.field_eval_rays(pf_f, rays, z_f, film)
        rgb_f, depth_f, acc_f, _ = ops.composite(raw_f, z_f, rays, want_weights=False)
        ctx.pf_c, ctx.pf_f, ctx.film = pf_c, pf_f, film
        ctx.n_c = len(pf_c.params)
        ctx.save_for_backward(rays, z_c, raw_c, z_f, raw_f)
        ctx.set_materialize_grads(False)
        return rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f

    @staticmethod
    def backward(ctx, g_rgb_c, g_depth_c, g_acc_c, g_rgb_f, g_depth_f, g_acc_f):
        rays, z_c, raw_c, z_f, raw_f = ctx.saved_tensors
        pf_c, pf_f = ctx.pf_c, ctx.pf_f
        same = pf_c is pf_f
        grads_c = grads_f = None
        if any(g is not None for g in (g_rgb_c, g_depth_c, g_acc_c)):
            g_raw = _composite_bwd(raw_c, z_c, rays, g_rgb_c, g_depth_c, g_acc_c)
            grads_c = _field_backward(pf_c, rays, z_c, raw_c, g_raw, ctx.film)
        if any(g is not None for g in (g_rgb_f, g_depth_f, g_acc_f)):
            g_raw = _composite_bwd(raw_f, z_f, rays, g_rgb_f, g_depth_f, g_acc_f)
            grads_f = _field_backward(pf_f, rays, z_f, raw_f, g_raw, ctx.film)
        if same:
            if grads_c is not None and grads_f is not None:
                torch._foreach_add_(grads_c, grads_f)
            merged = grads_c if grads_c is not None else grads_f
            param_grads = tuple(merged) if merged is not None else (None,) * ctx.n_c
        else:
            gc = tuple(grads_c) if grads_c is not None else (None,) * len(pf_c.params)
            gf = tuple(grads_f) if grads_f is not None else (None,) * len(pf_f.params)
            param_grads = gc + gf
        return (None,) * 10 + param_grads


def render_rays_train(pf_c, pf_f, rays, near, far, nc, nf, film, t_rand, seed):
    """render_rays with gradients to the field parameters (6-tuple like render.py:147)."""
    if film is not None:
        raise _lib.MiRenderError("training through FiLM fields is not implemented yet")
    params = list(pf_c.params) if pf_c is pf_f else list(pf_c.params) + list(pf_f.params)
    return _RenderRaysFn.apply(pf_c, pf_f, rays.detach(), float(near), float(far), int(nc), int(nf), film,
                               None if t_rand is None else t_rand.detach(), int(seed), *params)
