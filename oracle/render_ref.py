"""Oracle (test infrastructure): CPU restatement of the volume-render path.

Follows nerf/render.py:7-147 (byte-identical to pi_GAN/render.py:52-192) stage by
stage, in PyTorch-CPU fp32, with two deliberate interface differences that do not
change the arithmetic: the stratified jitter ``t_rand`` is an explicit argument
(the reference draws it from the global RNG at render.py:131), and every stage can
return its intermediates so each HIP kernel is checked with injected inputs.

Quirks kept on purpose: white background always added; last
delta = 1e10; +1e-10 inside the transmittance product; +1e-5 on pdf weights and
the denom<1e-5 guard; searchsorted(right=True); deterministic u; un-normalised
rays_d for points and delta scaling, normalised for the view input.
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import numpy as np
import torch

POINT_CHUNK = 65536   # run_network chunk, nerf/render.py:59
RAY_CHUNK = 16384     # render_image chunk, nerf/render.py:150


def get_rays(width, height, focal, c2w):
    """nerf/render.py:7-23: pinhole rays, NumPy, no pixel-centre offset, -z forward."""
    c2w = np.asarray(c2w)
    px, py = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32), indexing="xy")
    cam = np.stack([(px - width * 0.5) / focal, -(py - height * 0.5) / focal, -np.ones_like(px)], axis=-1)
    rays_d = (cam[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape)
    return rays_o, rays_d


def rays_from_camera(width, height, focal, c2w) -> np.ndarray:
    """[H*W, 2, 3] fp32 ray list in the order render_image builds it (render.py:151-154)."""
    o, d = get_rays(width, height, focal, c2w)
    return np.stack([o, d], axis=2).reshape(-1, 2, 3).astype(np.float32)


def stratified_z(n_rays: int, near: float, far: float, n_coarse: int, t_rand: torch.Tensor):
    """render.py:123-132.  Returns (z_vals[N,Nc], mids[N,Nc-1])."""
    z = torch.linspace(near, far, steps=n_coarse, dtype=t_rand.dtype).unsqueeze(0).expand(n_rays, n_coarse)
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    hi = torch.cat([mids, z[..., -1:]], -1)
    lo = torch.cat([z[..., :1], mids], -1)
    return lo + (hi - lo) * t_rand, mids


def points_on_rays(rays_o, rays_d, z):
    """render.py:134."""
    return rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]


def query_field(pts, view_dirs, field: Callable, chunk: int = POINT_CHUNK):
    """render.py:59-75: flatten, append per-ray view dir, chunked evaluation."""
    n, s = pts.shape[0], pts.shape[1]
    x = torch.cat([pts.reshape(-1, 3), view_dirs[:, None].expand(n, s, 3).reshape(-1, 3)], -1)
    out = torch.cat([field(x[i:i + chunk]) for i in range(0, x.shape[0], chunk)])
    return out.reshape(n, s, 4)


def composite(raw, z, rays_d):
    """render.py:78-103.  Returns (rgb[N,3], depth[N], acc[N], weights[N,S])."""
    n = raw.shape[0]
    delta = z[..., 1:] - z[..., :-1]
    delta = torch.cat([delta, torch.full((n, 1), 1e10, dtype=z.dtype)], -1)
    delta = delta * torch.norm(rays_d, dim=-1, keepdim=True)
    alpha = 1.0 - torch.exp(-raw[..., 3] * delta)
    trans = torch.cumprod(torch.cat([torch.ones((n, 1), dtype=z.dtype), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    rgb = torch.sum(w[..., None] * raw[..., :3], -2)
    depth = torch.sum(w * z, -1)
    acc = torch.sum(w, -1)
    rgb = rgb + (1.0 - acc[..., None])
    return rgb, depth, acc, w


def sample_pdf(bins, weights, n_samples: int):
    """render.py:27-56: inverse-CDF resampling with deterministic u."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.linspace(0.0, 1.0, steps=n_samples, dtype=cdf.dtype).expand(list(cdf.shape[:-1]) + [n_samples]).contiguous()
    idx = torch.searchsorted(cdf.detach(), u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


class RenderTrace(NamedTuple):
    rgb_c: torch.Tensor
    depth_c: torch.Tensor
    acc_c: torch.Tensor
    rgb_f: torch.Tensor
    depth_f: torch.Tensor
    acc_f: torch.Tensor
    z_coarse: torch.Tensor
    raw_c: torch.Tensor
    weights_c: torch.Tensor
    z_samples: torch.Tensor
    z_fine: torch.Tensor
    raw_f: torch.Tensor
    weights_f: torch.Tensor

    def outputs(self):
        return tuple(self[:6])


def render_rays(rays, near, far, coarse_field, fine_field, n_coarse, n_fine, t_rand,
                z_fine_override=None) -> RenderTrace:
    """render.py:106-147 with the jitter injected.  ``z_fine_override`` replaces the
    sorted fine depths (used to test the fine MLP+composite stage in isolation,
    because hierarchical resampling is ill-conditioned: SURVEY.md §8c)."""
    rays_o, rays_d = rays[:, 0], rays[:, 1]
    view = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    z_c, mids = stratified_z(rays.shape[0], near, far, n_coarse, t_rand)
    raw_c = query_field(points_on_rays(rays_o, rays_d, z_c), view, coarse_field)
    rgb_c, depth_c, acc_c, w_c = composite(raw_c, z_c, rays_d)

    z_s = sample_pdf(mids, w_c[..., 1:-1], n_fine).detach()
    if z_fine_override is None:
        z_f, _ = torch.sort(torch.cat([z_c, z_s], -1), -1)
    else:
        z_f = z_fine_override
    raw_f = query_field(points_on_rays(rays_o, rays_d, z_f), view, fine_field)
    rgb_f, depth_f, acc_f, w_f = composite(raw_f, z_f, rays_d)
    return RenderTrace(rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f, z_c, raw_c, w_c, z_s, z_f, raw_f, w_f)


def render_image(width, height, focal, pose, near, far, coarse_field, fine_field, n_coarse, n_fine,
                 t_rand, chunk: int = RAY_CHUNK):
    """render.py:150-167 (nerf flavour: numpy rgb[H,W,3], depth[H,W,1], acc[H,W,1]); jitter
    ``t_rand[H*W, Nc]`` is sliced per ray chunk."""
    rays = torch.from_numpy(rays_from_camera(width, height, focal, pose))
    parts = []
    for i in range(0, rays.shape[0], chunk):
        tr = render_rays(rays[i:i + chunk], near, far, coarse_field, fine_field, n_coarse, n_fine,
                         t_rand[i:i + chunk])
        parts.append((tr.rgb_f, tr.depth_f, tr.acc_f))
    rgb = torch.cat([p[0] for p in parts]).reshape(height, width, 3)
    depth = torch.cat([p[1] for p in parts]).reshape(height, width, 1)
    acc = torch.cat([p[2] for p in parts]).reshape(height, width, 1)
    return rgb.numpy(), depth.numpy(), acc.numpy()


def render_rays_f64(rays, near, far, coarse_field64, fine_field64, n_coarse, n_fine, t_rand,
                    z_fine_override=None) -> RenderTrace:
    """The same algorithm evaluated in fp64 (fields must hold fp64 weights).  Used by tests to measure
    the fp32 CPU path's own distance from exact arithmetic: a stage whose fp32 oracle already deviates by
    more than the 1e-4 gate from its fp64 self is judged against that floor, not against 1e-4."""
    z_o = None if z_fine_override is None else z_fine_override.double()
    return render_rays(rays.double(), near, far, coarse_field64, fine_field64, n_coarse, n_fine, t_rand.double(), z_o)


def psnr(a, b) -> float:
    """-10 log10(MSE), nerf/train_nerf.py:160."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else -10.0 * np.log10(mse)
