"""The reference's `render` call surface on the HIP path.

Mirrors, name for name and argument for argument, nerf/render.py (get_rays :7, sample_pdf :27,
run_network :59, raw_to_outputs :78, render_rays :106, render_image :150, render_video :170) and the
pi_GAN variants (pi_GAN/render.py:195-241).  Differences that do not change results:

* device comes from the inputs / the model, not from the global default tensor type;
* `render_rays` takes optional keyword-only `t_rand` / `seed` (the reference draws the jitter from
  the global RNG, render.py:131, which cannot be matched across devices);
* whole images are rendered in one launch sequence per (up to) 2^20 rays instead of 16 384-ray
  chunks with a device->host sync each (render.py:158-163); `chunk` is accepted and ignored
  unless smaller memory is needed.

Known field models (fields.py) run the fused MLP; any other callable `network([M,6]) -> [M,4]`
is driven through the same sampling / compositing kernels (generic path).
"""
from __future__ import annotations

import numpy as np
import torch
from tqdm import tqdm

from . import _lib, fields, ops

to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)  # noqa: E731  (nerf/render.py:5)

MAX_RAYS_PER_LAUNCH = 1 << 20


def _device_of(*models, rays=None):
    for m in models:
        pf = fields.as_packed_field(m) if isinstance(m, torch.nn.Module) else None
        if pf is not None:
            return pf.device
        if isinstance(m, torch.nn.Module):
            for p in m.parameters():
                if p.is_cuda:
                    return p.device
    if isinstance(rays, torch.Tensor) and rays.is_cuda:
        return rays.device
    if not torch.cuda.is_available():
        raise _lib.MiRenderError("no ROCm device: the renderer has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _fresh_seed() -> int:
    # one draw from torch's CPU generator: reproducible under torch.manual_seed, no device sync (device="cpu"
    # explicitly: under the scripts' CUDA default tensor type a bare randint is a device draw + sync)
    return int(torch.randint(0, 2 ** 62, (1,), device="cpu").item())


def get_rays(width, height, focal, c2w):
    """nerf/render.py:7-23.  Returns NumPy (rays_o, rays_d) [H,W,3] fp32 like the reference; the
    arithmetic runs in the HIP ray generator (bit-identical to the NumPy expression)."""
    dev = _device_of()
    rays = ops.gen_rays(width, height, focal, c2w, dev).cpu().numpy().reshape(height, width, 2, 3)
    return np.ascontiguousarray(rays[:, :, 0]), np.ascontiguousarray(rays[:, :, 1])


def sample_pdf(bins, weights, N_samples):
    """nerf/render.py:27-56: inverse-CDF samples [N, N_samples] from per-ray bins [N,nb] and weights [N,nb-1]."""
    dev = bins.device if isinstance(bins, torch.Tensor) and bins.is_cuda else _device_of()
    return ops.sample_pdf(torch.as_tensor(bins).to(dev), torch.as_tensor(weights).to(dev), int(N_samples))


def run_network(ray_samples, view_dirs, network, chunk=1024 * 64):
    """nerf/render.py:59-75: flatten, append per-ray view dirs, evaluate `network`."""
    n, s = ray_samples.shape[0], ray_samples.shape[1]
    x = torch.cat([ray_samples.reshape(-1, 3), view_dirs[:, None].expand(n, s, 3).reshape(-1, 3)], -1)
    pf = fields.as_packed_field(network)
    if pf is not None and not fields.is_film(pf.kind):
        from . import autograd
        out = autograd.field_eval_points(pf, x)            # one launch; gradients reach the parameters like the reference's
    else:
        out = torch.cat([network(x[i:i + chunk]) for i in range(0, x.shape[0], chunk)])
    return out.reshape(n, s, 4)


def raw_to_outputs(raw, z_vals, rays_d):
    """nerf/render.py:78-103 on the compositing kernel: (rgb, depth, acc, weights), differentiable with respect to `raw`
    like the reference's torch ops (backward: mi_composite_bwd).  `z_vals` / `rays_d` carry no gradient on the reference's
    path (render.py:141 detaches the resampled depths); passing ones that require grad raises instead of dropping it."""
    from . import autograd
    rays = torch.stack([torch.zeros_like(rays_d), rays_d], 1)
    return autograd.composite(raw, z_vals, rays, want_weights=True)


def _generic_field(network, rays, z, chunk=1024 * 64):
    """Unknown callable: points/view dirs are formed with torch ops on the device and fed to it; its own autograd graph
    (parameters, FiLM leaves, whatever it closes over) reaches the returned raw values."""
    o, d = rays[:, 0], rays[:, 1]
    view = d / torch.norm(d, dim=-1, keepdim=True)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    return run_network(pts, view, network, chunk).to(torch.float32).contiguous()


def _eval_pass(model, pf, rays, z, film):
    """raw [n,S,4] of one pass on the generic path: a fused kind (one side of a mixed pair) through its own kernels with
    gradients to its parameters / FiLM table, anything else through its own forward."""
    if pf is None:
        return _generic_field(model, rays, z)
    from . import autograd
    if fields.is_film(pf.kind) and film is None:
        film = fields.film_table(model)
    return autograd.field_eval_rays(pf, rays, z, film if fields.is_film(pf.kind) else None)


def render_rays(rays, near, far, coarse_model, fine_model, coarse_sample_num, fine_sample_num, *,
                t_rand=None, seed=None, film=None, ray0=0):
    """nerf/render.py:106-147.  rays [N,2,3] -> (rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f).
    `ray0`: index of rays[0] in the caller's full ray list; the seeded jitter is keyed by (seed, ray0 + k, sample),
    so a frame rendered in pieces (chunks, GPUs) gets the jitter it would get in one call.
    `film` [b,9,512] (FiLM fields only) renders b images in one call: rays are b equal consecutive groups and
    group g uses film[g]; by default the model's own film_params (one image) are used like the reference."""
    dev = _device_of(coarse_model, fine_model, rays=rays)
    rays = torch.as_tensor(rays).to(device=dev, dtype=torch.float32).reshape(-1, 2, 3).contiguous()
    nc, nf = int(coarse_sample_num), int(fine_sample_num)
    if seed is None and t_rand is None:
        seed = _fresh_seed()
    pf_c, pf_f = fields.as_packed_field(coarse_model), fields.as_packed_field(fine_model)
    if rays.shape[0] == 0:
        # no rays (an empty batch, a rank of a group larger than the frame): six empty outputs; under autograd they hang
        # off the parameters with zero gradients, so a loss over an empty shard still backpropagates
        outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in ((0, 3), (0,), (0,), (0, 3), (0,), (0,))]
        leaves = [p for m in (coarse_model, fine_model) if isinstance(m, torch.nn.Module) for p in m.parameters() if p.requires_grad]
        # ... and off the FiLM source (the mapping network's output, a GAN-inversion leaf): a rank with an empty shard must
        # hand allreduce_grads the same set of gradients as every other rank
        for m in (coarse_model, fine_model):
            fp = film if film is not None else getattr(m, "film_params", None)
            for t in ([fp] if isinstance(fp, torch.Tensor) else [x for pair in (fp or []) for x in pair]):
                if isinstance(t, torch.Tensor) and t.requires_grad and not any(t is u for u in leaves):
                    leaves.append(t)
        if torch.is_grad_enabled() and leaves:
            zero = sum(p.reshape(-1)[:1].sum() for p in leaves) * 0.0
            outs = [o + zero for o in outs]
        return tuple(outs)
    if pf_c is not None and pf_f is not None:
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in pf_c.params + pf_f.params)
        if fields.is_film(pf_c.kind) or fields.is_film(pf_f.kind):
            if film is None:
                film = fields.film_table(coarse_model if fields.is_film(pf_c.kind) else fine_model)
            needs_grad = needs_grad or (torch.is_grad_enabled() and film.requires_grad)
        else:
            film = None
        if needs_grad:
            from . import autograd
            return autograd.render_rays_train(pf_c, pf_f, rays, near, far, nc, nf, film, t_rand, seed or 0, ray0)
        return ops.render_rays_fused(pf_c, pf_f, rays, near, far, nc, nf, film, t_rand, seed or 0, ray0=ray0)

    # generic path: sampling / compositing kernels around an arbitrary callable (or a mixed pair: one fused kind, one
    # callable).  Differentiable like the reference's torch ops (render.py:59-103): the callable's graph reaches `raw`,
    # compositing continues it (autograd.composite -> mi_composite_bwd), a fused side goes through its own backward
    # kernels (autograd.field_eval_rays); the resampled depths are detached as at render.py:141.
    from . import autograd
    n = rays.shape[0]
    z_c = ops.sample_coarse(n, near, far, nc, dev, t_rand, seed or 0, ray0=ray0)
    raw_c = _eval_pass(coarse_model, pf_c, rays, z_c, film)
    rgb_c, depth_c, acc_c, w_c = autograd.composite(raw_c, z_c, rays)
    z_f = ops.sample_fine(z_c, w_c.detach(), near, far, nf)
    raw_f = _eval_pass(fine_model, pf_f, rays, z_f, film)
    rgb_f, depth_f, acc_f, _ = autograd.composite(raw_f, z_f, rays, want_weights=False)
    return rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f


def _render_image_device(width, height, focal, pose, near, far, coarse_model, fine_model, nc, nf, chunk,
                         t_rand=None, seed=None, ray0=0, n_rays=None):
    """Rays [ray0, ray0+n) of the image -> device tensors (rgb[n,3], depth[n], acc[n]) of the FINE pass."""
    dev = _device_of(coarse_model, fine_model)
    total = width * height
    n_rays = total - ray0 if n_rays is None else n_rays
    step = MAX_RAYS_PER_LAUNCH if not chunk else max(int(chunk), 2)
    step = max(step, 16384)
    if seed is None and t_rand is None:
        seed = _fresh_seed()
    parts = []
    for i in range(ray0, ray0 + n_rays, step):
        m = min(step, ray0 + n_rays - i)
        rays = ops.gen_rays(width, height, focal, pose, dev, i, m)
        tr = None if t_rand is None else t_rand[i - ray0:i - ray0 + m]   # row k of t_rand <-> ray ray0 + k
        out = render_rays(rays, near, far, coarse_model, fine_model, nc, nf, t_rand=tr, seed=seed, ray0=i)
        parts.append(out[3:6])
    if not parts:                                          # an empty ray range (a rank of a group larger than the frame)
        return (torch.empty((0, 3), dtype=torch.float32, device=dev), torch.empty(0, dtype=torch.float32, device=dev),
                torch.empty(0, dtype=torch.float32, device=dev))
    if len(parts) == 1:
        return parts[0]
    return tuple(torch.cat([p[k] for p in parts]) for k in range(3))


def render_image(width, height, focal, pose, near, far, coarse_model, fine_model, coarse_sample_num,
                 fine_sample_num, chunk=1024 * 16, *, t_rand=None, seed=None):
    """nerf/render.py:150-167: NumPy rgb[H,W,3], depth[H,W,1], acc[H,W,1] of the fine pass."""
    with torch.no_grad():
        rgb, depth, acc = _render_image_device(width, height, focal, pose, near, far, coarse_model, fine_model,
                                               int(coarse_sample_num), int(fine_sample_num), None, t_rand, seed)
    return (rgb.cpu().numpy().reshape(height, width, 3), depth.cpu().numpy().reshape(height, width, 1),
            acc.cpu().numpy().reshape(height, width, 1))


render_image_np = render_image   # pi_GAN/render.py:209-226 is the same function


def render_image_tensor(width, height, focal, pose, near, far, coarse_model, fine_model, coarse_sample_num,
                        fine_sample_num, chunk=1024 * 16, *, t_rand=None, seed=None):
    """pi_GAN/render.py:195-206: rgb of the fine pass as a device tensor [H,W,3] carrying the autograd graph."""
    rgb, _, _ = _render_image_device(width, height, focal, pose, near, far, coarse_model, fine_model,
                                     int(coarse_sample_num), int(fine_sample_num), None, t_rand, seed)
    return rgb.reshape(height, width, 3)


def render_video(width, height, focal, poses, near, far, coarse_model, fine_model, coarse_sample_num,
                 fine_sample_num, chunk=1024 * 16, *, t_rand=None, seed=None):
    """nerf/render.py:170-182 (and the evident intent of pi_GAN/render.py:229-241, whose body unpacks
    three values from the tensor-returning render_image): stacked NumPy frames rgb[F,H,W,3], depth[F,H,W,1],
    acc[F,H,W,1].  Keyword-only extras: `t_rand` [F, H*W, Nc] injects each frame's jitter, `seed` seeds frame i with
    seed + i (the reference draws every frame's jitter from the global RNG)."""
    # Frame i's three device->host copies run on a side stream into pinned memory while frame i + 1 renders (the
    # reference - and render_image - stop for three pageable copies after every frame); at most two frames in flight.
    poses = list(poses)
    n, nframes = int(width) * int(height), len(poses)
    if nframes == 0:
        return (np.zeros((0, height, width, 3), np.float32), np.zeros((0, height, width, 1), np.float32),
                np.zeros((0, height, width, 1), np.float32))
    pin = nframes * n * 20 <= (2 << 30)        # page-locking more than 2 GiB for one call is not ours to decide
    try:
        host = [torch.empty((nframes, n, c), dtype=torch.float32, device="cpu", pin_memory=pin) for c in (3, 1, 1)]
    except RuntimeError:                       # no pinned memory to be had: pageable copies, still off the render stream
        host = [torch.empty((nframes, n, c), dtype=torch.float32, device="cpu") for c in (3, 1, 1)]
    copy_stream, done = None, []
    for i, p in enumerate(tqdm(poses)):
        with torch.no_grad():
            outs = _render_image_device(width, height, focal, p, near, far, coarse_model, fine_model,
                                        int(coarse_sample_num), int(fine_sample_num), None,
                                        None if t_rand is None else t_rand[i], None if seed is None else seed + i)
        dev = outs[0].device
        if copy_stream is None:
            copy_stream = torch.cuda.Stream(dev)
        rendered = torch.cuda.Event()
        rendered.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(rendered)
            for h, t in zip(host, outs):
                h[i].copy_(t.reshape(n, -1), non_blocking=True)
                t.record_stream(copy_stream)   # the allocator must not hand the frame out again before its copy ran
            copied = torch.cuda.Event()
            copied.record(copy_stream)
        done.append(copied)
        if len(done) > 1:
            done.pop(0).synchronize()
    copy_stream.synchronize()
    return (host[0].numpy().reshape(nframes, height, width, 3), host[1].numpy().reshape(nframes, height, width, 1),
            host[2].numpy().reshape(nframes, height, width, 1))


render_video_np = render_video
