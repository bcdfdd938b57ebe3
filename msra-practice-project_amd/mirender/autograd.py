"""Differentiable render_rays: what PyTorch autograd provides in the reference (nerf/train_nerf.py:151-168,
pi_GAN/modules.py:159-161), rebuilt on the HIP kernels.

Forward = the training kernels, which also keep every layer's input, for as many ray ranges as the memory budget
holds, and the inference kernels for the rest.  Backward, per pass that received a gradient: compositing
backward -> dL/d(raw); then range by range (so memory stays bounded whatever the batch): the field forward is
re-run if the range was not kept, the backward chain produces every layer's dA, and the point-contraction GEMMs
reduce them to the weight gradients (csrc/field_mlp_bwd.hip).  z_samples is detached like the reference (render.py:141), so the fine depths
carry no gradient into the coarse pass.
"""
from __future__ import annotations

import ctypes
import os
import sys

import torch

from . import _lib, fields, ops

CHUNK_BYTES = 48 << 30         # saved activations + per-layer gradients per ray range (288 GB of HBM)
SAVE_COARSE_BYTES = 48 << 30   # layer inputs the forward may keep for the coarse pass (no recompute in backward) ...
SAVE_FINE_BYTES = 208 << 30    # ... and for the fine pass; both also limited to the free memory minus RESERVE_BYTES
RESERVE_FIXED_BYTES = 16 << 30   # what backward needs besides the kept inputs and one range's rows: reduction scratch,
#                                  raw / depths / their gradients, the allocator's rounding


# MI_DEBUG_GUARDS=1 (tests/test_gpu_guards.py): every scratch / row buffer handed to the library gets GUARD extra floats of a
# sentinel at its end, checked after the call - the library never allocates, so a kernel writing past what a *_floats()
# query promised would otherwise corrupt a neighbouring tensor silently.
GUARD = 4096
_SENTINEL = 12345.678


def _guarded(n: int, dev):
    """torch.empty(n) fp32 on dev; with MI_DEBUG_GUARDS=1: (n + GUARD) with the tail set to the sentinel."""
    import os
    if os.environ.get("MI_DEBUG_GUARDS") != "1":
        return torch.empty(int(n), dtype=torch.float32, device=dev), None
    buf = torch.empty(int(n) + GUARD, dtype=torch.float32, device=dev)
    buf[int(n):] = _SENTINEL
    return buf[:int(n)], buf


def _check_guard(whole, what: str):
    if whole is not None and not bool((whole[-GUARD:] == _SENTINEL).all()):
        bad = int((whole[-GUARD:] != _SENTINEL).nonzero()[-1]) + 1
        raise _lib.MiRenderError(f"{what}: the library wrote {bad} floats past the end of the buffer")


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


def _shape(rays, z):
    """(units, samples per unit) of a pass.  Two input forms share every function below: points o + d z on rays
    (`rays` [n,2,3], `z` [n,S]: run_network inside render_rays, render.py:134-135) and free-standing points (`rays` holds
    x [M,6] = position | view direction, `z` is None: `network(x)` called on its own, render.py:73) - then a unit is a point."""
    return (rays.shape[0], 1) if z is None else tuple(z.shape)


def _cut(z, r0, r1):
    return None if z is None else z[r0:r1]


def _forward_plain(pf: fields.PackedField, rays, z, film):
    if z is None:
        return fields.eval_points(pf, rays, film).reshape(-1, 1, 4)
    return ops.field_eval_rays(pf, rays, z, film)


def _forward_saving(pf: fields.PackedField, rays, z, film):
    """Field forward that also keeps every layer's input (training forward).  Returns (raw, acts)."""
    lib = _lib.load()
    dev = pf.device
    n, s = _shape(rays, z)
    pts = n * s
    f, groups, rpg = _groups(pf, film, n)
    acts, acts_g = _guarded(lib.mi_field_train_acts_floats(pf.kind) * pts, dev)
    raw = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if z is None:
            _lib.check(lib.mi_field_eval_points_train(pf.kind, _lib.ptr(pf.refresh()), _lib.ptr(f), _lib.ptr(rays), groups, rpg,
                                                      _lib.ptr(raw), _lib.ptr(acts), _lib.stream_ptr(dev)),
                       "mi_field_eval_points_train")
        else:
            _lib.check(lib.mi_field_eval_rays_train(pf.kind, _lib.ptr(pf.refresh()), _lib.ptr(f), _lib.ptr(rays), _lib.ptr(z),
                                                    groups, rpg, s, _lib.ptr(raw), _lib.ptr(acts), _lib.stream_ptr(dev)),
                       "mi_field_eval_rays_train")
    _check_guard(acts_g, "saved layer inputs (mi_field_train_acts_floats)")
    return raw, acts


def _chunk_ranges(pf: fields.PackedField, n: int, s: int, film):
    """The pass split into ray ranges whose saved activations + gradients fit CHUNK_BYTES (whole FiLM groups per
    range).  Forward and backward use the same split, so a range saved in the forward is found again."""
    f_all, groups, rpg = _groups(pf, film, n)
    max_rays = max(1, _max_points_per_chunk(pf) // s)
    if f_all is None or max_rays >= rpg:
        step = max_rays if f_all is None else (max_rays // rpg) * rpg          # whole FiLM groups per range
        return f_all, rpg, [(r0, min(n, r0 + step)) for r0 in range(0, n, step)]
    # one image alone exceeds the budget (e.g. 256x256 at 72 samples): equal parts of one group per range
    parts = -(-rpg // max_rays)
    step = -(-rpg // parts)
    return f_all, rpg, [(g * rpg + a, g * rpg + min(rpg, a + step)) for g in range(groups) for a in range(0, rpg, step)]


def _film_of_range(f_all, rpg, r0, r1):
    """FiLM rows of the range: its whole groups, or the one group it is a part of."""
    if f_all is None:
        return None
    return f_all[r0 // rpg:max(r0 // rpg + 1, r1 // rpg)]


def _save_budget(dev, cap: int, range_bytes: int) -> int:
    """Bytes of layer inputs the forward may keep: `cap`, but never more than what can really be had right now minus
    what backward will ask for on top - one range's recomputed inputs + per-layer gradients (`range_bytes`, the largest
    range of THIS pass, not the CHUNK_BYTES ceiling) + RESERVE_FIXED_BYTES.  "Can be had" = the driver's free figure
    plus the part of torch's cache that is whole free segments; the unused remainders of partly used segments
    (`inactive_split_bytes`) cannot be handed back to the driver or merged, so they do not count."""
    free, _total = torch.cuda.mem_get_info(dev)
    stats = torch.cuda.memory_stats(dev)
    cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    cached -= stats.get("inactive_split_bytes.all.current", 0)
    return max(0, min(cap, free + max(0, cached) - range_bytes - RESERVE_FIXED_BYTES))


def _forward_pass(pf: fields.PackedField, rays, z, film, cap: int, all_or_nothing: bool = False):
    """raw [n,S,4] of the pass, plus {range index: saved layer inputs} for as many leading ranges as the budget
    holds (with all_or_nothing: for every range or for none); the rest is evaluated by the plain kernel and
    recomputed range by range in backward."""
    lib = _lib.load()
    n, s = _shape(rays, z)
    per_point = 4 * lib.mi_field_train_acts_floats(pf.kind)
    f_all, rpg, ranges = _chunk_ranges(pf, n, s, film)
    per_point_bwd = per_point + 4 * lib.mi_field_train_grads_floats(pf.kind)
    budget = _save_budget(pf.device, cap, per_point_bwd * s * max(r1 - r0 for r0, r1 in ranges))
    if all_or_nothing and per_point * n * s > budget:
        budget = 0
    saved, parts, r_done = {}, [], 0
    for k, (r0, r1) in enumerate(ranges):
        need = per_point * (r1 - r0) * s
        if need > budget:
            break
        raw_k, saved[k] = _forward_saving(pf, rays[r0:r1], _cut(z, r0, r1), _film_of_range(f_all, rpg, r0, r1))
        parts.append(raw_k)
        budget -= need
        r_done = r1
    if os.environ.get("MI_DEBUG_PLAN") == "1":            # what the memory planner decided for this pass
        free, _t = torch.cuda.mem_get_info(pf.device)
        sys.stderr.write(f"[mirender plan] {n} rays x {s}: kept {len(saved)} of {len(ranges)} ranges ({r_done} rays), "
                         f"budget left {budget / 2**30:.1f} GiB, driver free {free / 2**30:.1f} GiB, torch reserved "
                         f"{torch.cuda.memory_reserved(pf.device) / 2**30:.1f} allocated {torch.cuda.memory_allocated(pf.device) / 2**30:.1f} GiB\n")
    if r_done < n:
        if r_done % rpg:                                   # finish the image the kept ranges stopped inside
            r_next = (r_done // rpg + 1) * rpg
            parts.append(_forward_plain(pf, rays[r_done:r_next], _cut(z, r_done, r_next), _film_of_range(f_all, rpg, r_done, r_next)))
            r_done = r_next
        if r_done < n:
            parts.append(_forward_plain(pf, rays[r_done:], _cut(z, r_done, n), None if f_all is None else f_all[r_done // rpg:]))
    return (parts[0] if len(parts) == 1 else torch.cat(parts)), saved


def _field_backward(pf: fields.PackedField, rays, z, raw, g_raw, film, saved=None):
    """Gradients of one pass: (list of tensors shaped like pf.params, grad of the FiLM table or None).
    `saved` = {range index: layer inputs kept by the forward}; the other ranges re-run the forward first, so
    memory stays bounded whatever the batch."""
    lib = _lib.load()
    dev = pf.device
    grads_f = lib.mi_field_train_grads_floats(pf.kind)
    n, s = _shape(rays, z)
    f_all, rpg, ranges = _chunk_ranges(pf, n, s, film)
    saved = {} if saved is None else saved
    total = None
    g_film = None if f_all is None else torch.empty_like(f_all)
    packed_bwd = pf.refresh_bwd()
    stream = _lib.stream_ptr(dev)
    def one_range(k, r0, r1):
        pts = (r1 - r0) * s
        f_c = g_c = fp = None
        ng, ppg = 1, pts
        add_to_row, g0 = False, 0
        if f_all is not None:
            f_c = _film_of_range(f_all, rpg, r0, r1)
            ng = f_c.shape[0]
            ppg = pts // ng                                   # whole groups, or this part of one group
            g0 = r0 // rpg
            part_of_group = (r1 - r0) < rpg
            # a part of an image adds to that image's row; the first part (and whole groups) overwrite
            add_to_row = part_of_group and r0 % rpg != 0
            g_c = torch.empty_like(f_c) if add_to_row else g_film[g0:g0 + ng]
            fp = torch.empty(lib.mi_field_film_partial_floats(ng, ppg), dtype=torch.float32, device=dev)
        gws, gws_g = _guarded(grads_f * pts, dev)
        part, part_g = _guarded(lib.mi_field_bwd_partial_floats(pts), dev)
        out = [torch.empty_like(p) for p in pf.params]
        if k in saved:
            acts_c, raw_c = saved.pop(k), raw[r0:r1]
        else:
            raw_c, acts_c = _forward_saving(pf, rays[r0:r1], _cut(z, r0, r1), f_c)
        arr = (ctypes.c_void_p * len(out))(*[t.data_ptr() for t in out])
        # FiLM kinds: d gamma = <W, dW_image> + b . db_image needs the parameters themselves
        par = (ctypes.c_void_p * len(out))(*[p.data_ptr() for p in pf.params]) if f_all is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.mi_field_backward(pf.kind, _lib.ptr(packed_bwd), _lib.ptr(f_c), _lib.ptr(acts_c), _lib.ptr(gws),
                                             _lib.ptr(raw_c), _lib.ptr(g_raw[r0:r1]), ng, ppg, _lib.ptr(part),
                                             _lib.ptr(fp), arr, par, len(out), _lib.ptr(g_c), stream), "mi_field_backward")
        del acts_c
        _check_guard(gws_g, "per-layer gradients (mi_field_train_grads_floats)")
        _check_guard(part_g, "backward scratch (mi_field_bwd_partial_floats)")
        if f_all is not None and add_to_row:
            g_film[g0:g0 + 1] += g_c
        return out

    for k, (r0, r1) in enumerate(ranges):
        retry = False
        try:
            out = one_range(k, r0, r1)
        except torch.cuda.OutOfMemoryError:
            # The forward kept more layer inputs than this backward can live next to (its budget is an estimate of what
            # the allocator can still give).  Nothing of range k has been launched yet - allocations come first in
            # one_range - so give the kept inputs of the ranges still ahead back, and recompute them range by range:
            # same gradients bit for bit (tests/test_gpu_train.py), bounded memory.
            if not saved:
                raise
            retry = True
        if retry:
            # outside the handler: while it runs, the exception's traceback keeps the failed call's frame - and with it
            # whatever that call had already allocated (gradient rows, scratch) - alive, and empty_cache() could not
            # hand those blocks back exactly when the retry needs them
            saved.clear()
            torch.cuda.empty_cache()
            out = one_range(k, r0, r1)
        if total is None:
            total = out
        else:
            torch._foreach_add_(total, out)
    return total, g_film


def _composite_bwd(raw, z, rays, g_rgb, g_depth, g_acc, g_w=None):
    lib = _lib.load()
    dev = raw.device
    n, s = z.shape
    g_raw = torch.empty_like(raw)
    c = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    g_rgb, g_depth, g_acc, g_w = c(g_rgb), c(g_depth), c(g_acc), c(g_w)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_composite_bwd(n, s, _lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays), _lib.ptr(g_rgb),
                                        _lib.ptr(g_depth), _lib.ptr(g_acc), _lib.ptr(g_w), _lib.ptr(g_raw),
                                        _lib.stream_ptr(dev)), "mi_composite_bwd")
    return g_raw


class _RenderRaysFn(torch.autograd.Function):
    """render_rays with autograd.  Two shapes:

    * two fields (nerf's coarse / fine pair): coarse pass, resampling, fine pass over all Nc + Nf depths;
    * ONE field for both passes (pi_GAN: `render_image(..., model, model, ...)`, pi_GAN/modules.py:160-161; nerf with
      use_fine_model off, train_nerf.py:91,94): Nc of the fine pass's Nc + Nf points are the coarse pass's points, evaluated by
      the same field - identical inputs, identical outputs (the reference evaluates them twice, render.py:135,144).  The
      field runs on the Nf NEW depths only, `mi_merge_raw` puts both sets of raw values into the sorted order, and in
      backward `mi_split_grad` hands the fine composite's gradient back to the two point sets: the coarse points get the
      sum of what the coarse outputs and the fine outputs send them, and each set goes through the field's backward once.
      48 -> 36 field evaluations per ray forward at 12+24 (pi_GAN C4), 120 -> 108 evaluation-equivalents per training step;
      the outputs are bit-identical to evaluating all Nc + Nf points (tests/test_gpu_shared_field.py), the gradients equal
      up to the order of two partial sums."""

    @staticmethod
    def forward(ctx, pf_c, pf_f, rays, near, far, nc, nf, film, t_rand, seed, ray0, *params):
        dev = pf_c.device
        n = rays.shape[0]
        shared = pf_c is pf_f
        z_c = ops.sample_coarse(n, near, far, nc, dev, t_rand, seed, ray0=ray0)
        # two fields: the coarse pass often gets no gradient at all and keeps its layer inputs only when they are small.
        # One field: its points always take part in backward (through the fine composite): kept like the fine pass's
        raw_c, ctx.acts_c = _forward_pass(pf_c, rays, z_c, film, SAVE_FINE_BYTES if shared else SAVE_COARSE_BYTES,
                                          all_or_nothing=not shared)
        rgb_c, depth_c, acc_c, w_c = ops.composite(raw_c, z_c, rays)
        empty = torch.empty(0, device=dev)
        z_s = raw_s = pos = empty
        if not shared:
            z_f = ops.sample_fine(z_c, w_c, near, far, nf)
            raw_f, ctx.acts_f = _forward_pass(pf_f, rays, z_f, film, SAVE_FINE_BYTES)
        elif nf > 0:
            z_f, z_s, pos = ops.sample_fine_pos(z_c, w_c, near, far, nf)
            raw_s, ctx.acts_f = _forward_pass(pf_f, rays, z_s, film, SAVE_FINE_BYTES)        # the Nf new depths only
            raw_f = ops.merge_raw(raw_c, raw_s, pos)
        else:                                   # Nf = 0: sort(z_coarse) is z_coarse, the fine pass IS the coarse pass
            z_f, raw_f, ctx.acts_f = z_c, raw_c, {}
        rgb_f, depth_f, acc_f, _ = ops.composite(raw_f, z_f, rays, want_weights=False)
        ctx.pf_c, ctx.pf_f, ctx.film = pf_c, pf_f, None if film is None else film.detach()
        ctx.versions = (pf_c.versions(), pf_f.versions())
        ctx.n_c, ctx.shared, ctx.nc, ctx.nf = len(pf_c.params), shared, nc, nf
        ctx.save_for_backward(rays, z_c, raw_c, z_f, raw_f, z_s, raw_s, pos)
        ctx.set_materialize_grads(False)
        return rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f

    @staticmethod
    def backward(ctx, g_rgb_c, g_depth_c, g_acc_c, g_rgb_f, g_depth_f, g_acc_f):
        rays, z_c, raw_c, z_f, raw_f, z_s, raw_s, pos = ctx.saved_tensors
        pf_c, pf_f = ctx.pf_c, ctx.pf_f
        if (pf_c.versions(), pf_f.versions()) != ctx.versions:
            # backward re-reads the live weights (transposed stream, recomputed ranges) while raw and the kept layer
            # inputs date from the forward: a parameter updated in between would give silently mixed gradients.
            # PyTorch raises for its own saved tensors in this situation; so do we.
            raise RuntimeError("render_rays backward: a field parameter was modified in place (or replaced) after the "
                               "forward pass that this backward belongs to")
        grads_c = grads_f = gfilm_c = gfilm_f = None
        want_c = any(g is not None for g in (g_rgb_c, g_depth_c, g_acc_c))
        want_f = any(g is not None for g in (g_rgb_f, g_depth_f, g_acc_f))
        if not ctx.shared:
            if want_c:
                g_raw = _composite_bwd(raw_c, z_c, rays, g_rgb_c, g_depth_c, g_acc_c)
                grads_c, gfilm_c = _field_backward(pf_c, rays, z_c, raw_c, g_raw, ctx.film, ctx.acts_c)
            if want_f:
                g_raw = _composite_bwd(raw_f, z_f, rays, g_rgb_f, g_depth_f, g_acc_f)
                grads_f, gfilm_f = _field_backward(pf_f, rays, z_f, raw_f, g_raw, ctx.film, ctx.acts_f)
        else:
            g_c = _composite_bwd(raw_c, z_c, rays, g_rgb_c, g_depth_c, g_acc_c) if want_c else None
            g_s = None
            if want_f:
                g_f = _composite_bwd(raw_f, z_f, rays, g_rgb_f, g_depth_f, g_acc_f)
                if ctx.nf > 0:
                    g_c, g_s = ops.split_grad(g_f, pos, ctx.nc, g_c)      # onto the coarse outputs' own gradient, if any
                else:
                    g_c = g_f if g_c is None else g_c.add_(g_f)
                del g_f
            if g_c is not None:
                grads_c, gfilm_c = _field_backward(pf_c, rays, z_c, raw_c, g_c, ctx.film, ctx.acts_c)
            if g_s is not None:
                grads_f, gfilm_f = _field_backward(pf_f, rays, z_s, raw_s, g_s, ctx.film, ctx.acts_f)
        ctx.acts_c = ctx.acts_f = None          # release the saved activations
        if pf_c is pf_f:
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
        return (None,) * 7 + (g_film, None, None, None) + param_grads


def render_rays_train(pf_c, pf_f, rays, near, far, nc, nf, film, t_rand, seed, ray0=0):
    """render_rays with gradients to the field parameters and the FiLM table (6-tuple like render.py:147)."""
    params = list(pf_c.params) if pf_c is pf_f else list(pf_c.params) + list(pf_f.params)
    return _RenderRaysFn.apply(pf_c, pf_f, rays.detach(), float(near), float(far), int(nc), int(nf), film,
                               None if t_rand is None else t_rand.detach(), int(seed), int(ray0), *params)


# ----------------------------------------------------------------------------------------------------------------
# The stages on their own, differentiable: what the reference's plain torch ops give every caller of raw_to_outputs /
# run_network / network(x) (nerf/render.py:59-103), and what render_rays needs when a model is NOT one of the fused
# kinds (any callable, a w_0 != 30 look-alike, a mixed pair): the callable's own autograd graph reaches `raw`, and
# compositing / the fused side of a mixed pair continue it on the HIP kernels.
# ----------------------------------------------------------------------------------------------------------------
def _no_grad_input(t, what: str):
    if isinstance(t, torch.Tensor) and t.requires_grad and torch.is_grad_enabled():
        raise _lib.MiRenderError(
            f"{what} requires grad, but the compositing / field kernels produce no gradient for it: on the reference's path "
            "depths and rays never carry one (the jitter is data, z_samples is detached at nerf/render.py:141)")


class _CompositeFn(torch.autograd.Function):
    """raw_to_outputs (nerf/render.py:78-103): forward mi_composite, backward mi_composite_bwd -> dL/d(raw)."""

    @staticmethod
    def forward(ctx, raw, z, rays):
        rgb, depth, acc, w = ops.composite(raw, z, rays, want_weights=True)
        ctx.save_for_backward(raw.detach(), z.detach(), rays.detach())
        ctx.set_materialize_grads(False)
        return rgb, depth, acc, w

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_w):
        raw, z, rays = ctx.saved_tensors
        if all(g is None for g in (g_rgb, g_depth, g_acc, g_w)):
            return None, None, None
        dev = raw.device
        raw32, z32, rays32 = (t.to(device=dev, dtype=torch.float32).contiguous() for t in (raw, z, rays))
        return _composite_bwd(raw32, z32, rays32, g_rgb, g_depth, g_acc, g_w).to(raw.dtype).reshape(raw.shape), None, None


def composite(raw, z, rays, want_weights: bool = True):
    """ops.composite with autograd history towards `raw` (rgb[n,3], depth[n], acc[n], weights[n,S]|None)."""
    _no_grad_input(z, "z_vals")
    _no_grad_input(rays, "rays")
    if not (torch.is_grad_enabled() and isinstance(raw, torch.Tensor) and raw.requires_grad):
        return ops.composite(raw, z, rays, want_weights)
    rgb, depth, acc, w = _CompositeFn.apply(raw, z, rays)
    return rgb, depth, acc, (w if want_weights else None)


class _FieldFn(torch.autograd.Function):
    """network(inputs) of ONE pass for a fused kind - on rays (`z` [n,S]) or on free-standing points (`z` None, `rays` =
    x [M,6]) - with gradients to its parameters and to the FiLM table: the saving forward / backward chain / dW GEMMs of
    _RenderRaysFn, one pass at a time (a mixed pair's fused side; a fused module called on its own)."""

    @staticmethod
    def forward(ctx, pf, rays, z, film, *params):
        raw, ctx.acts = _forward_pass(pf, rays, z, film, SAVE_FINE_BYTES)
        ctx.pf, ctx.film, ctx.versions, ctx.points = pf, None if film is None else film.detach(), pf.versions(), z is None
        ctx.save_for_backward(rays, raw) if z is None else ctx.save_for_backward(rays, raw, z)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        rays, raw = ctx.saved_tensors[:2]
        z = None if ctx.points else ctx.saved_tensors[2]
        pf = ctx.pf
        if pf.versions() != ctx.versions:
            raise RuntimeError("field backward: a field parameter was modified in place (or replaced) after the forward "
                               "pass that this backward belongs to")
        grads, g_film = _field_backward(pf, rays, z, raw, g_raw.to(torch.float32).contiguous(), ctx.film, ctx.acts)
        ctx.acts = None
        if g_film is not None:
            g_film = g_film.reshape(ctx.film.shape) if ctx.needs_input_grad[3] else None
        return (None, None, None, g_film) + tuple(grads)


def _field_wants_grad(pf, film) -> bool:
    return torch.is_grad_enabled() and (any(p.requires_grad for p in pf.params) or
                                        (isinstance(film, torch.Tensor) and film.requires_grad))


def field_eval_rays(pf, rays, z, film=None):
    """ops.field_eval_rays (raw [n,S,4]) with autograd history towards the field's parameters and the FiLM table."""
    _no_grad_input(z, "z_vals")
    _no_grad_input(rays, "rays")
    if not _field_wants_grad(pf, film):
        return ops.field_eval_rays(pf, rays, z, film)
    dev = pf.device
    rays = rays.detach().to(device=dev, dtype=torch.float32).contiguous()
    z = z.detach().to(device=dev, dtype=torch.float32).contiguous()
    return _FieldFn.apply(pf, rays, z, film if fields.is_film(pf.kind) else None, *pf.params)


def field_eval_points(pf, x, film=None):
    """fields.eval_points (network(x [M,6]) -> [M,4], nerf/render.py:73) with autograd history towards the parameters and
    the FiLM table.  No gradient with respect to x: nothing on the reference's path asks for one."""
    _no_grad_input(x, "the field's input points")
    if not _field_wants_grad(pf, film):
        return fields.eval_points(pf, x, film)
    if x.dim() != 2 or x.shape[1] != 6:
        raise _lib.MiRenderError(f"expected inputs [M,6], got {tuple(x.shape)}")
    x = x.detach().to(device=pf.device, dtype=torch.float32).contiguous()
    if x.shape[0] == 0:
        return fields.eval_points(pf, x, film) + sum(p.reshape(-1)[:1].sum() for p in pf.params if p.requires_grad) * 0.0
    return _FieldFn.apply(pf, x, None, film if fields.is_film(pf.kind) else None, *pf.params).reshape(-1, 4)
