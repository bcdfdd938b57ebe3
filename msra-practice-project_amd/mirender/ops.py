"""Stage-level wrappers over the C ABI (one function per reference stage).

Every function takes/returns torch tensors on a ROCm device and launches on the calling
thread's current stream; nothing here computes in PyTorch.
"""
from __future__ import annotations

import ctypes
import functools
import os

import numpy as np
import torch

from . import _lib
from .fields import PackedField, is_film


def _f32c(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


@functools.lru_cache(maxsize=64)
def _linspace_table(start: float, end: float, steps: int, device_str: str):
    """torch.linspace evaluated on the CPU (as the reference's CPU path does: render.py:35,123) and
    uploaded once, so depths match the CPU oracle bit for bit."""
    # device="cpu" explicitly: the reference scripts set the CUDA default tensor type (nerf/train_nerf.py:11), under
    # which a bare factory call would evaluate the table with the device's linspace instead
    return torch.linspace(start, end, steps=steps, dtype=torch.float32, device="cpu").to(device_str)


def linspace_table(start, end, steps, device):
    if steps <= 0:
        return None
    return _linspace_table(float(start), float(end), int(steps), str(device))


def gen_rays(width: int, height: int, focal, c2w, device, ray0: int = 0, n: int | None = None) -> torch.Tensor:
    """get_rays (nerf/render.py:7-23) generated on the device in render_image's flattened order
    (render.py:151-154).  Returns rays [n,2,3]."""
    lib = _lib.load()
    n = width * height - ray0 if n is None else n
    c2w = np.ascontiguousarray(np.asarray(c2w)[:3, :4], dtype=np.float32)
    # NumPy >= 2 promotion: an np.float64 focal promotes the whole expression to fp64 (pi_GAN/modules.py:127),
    # a Python float keeps it fp32 (nerf/show_nerf.py:16)
    f64 = isinstance(focal, np.floating) and np.dtype(type(focal)) == np.float64
    rays = torch.empty((n, 2, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.mi_gen_rays(int(width), int(height), float(focal),
                                   c2w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(ray0), int(n),
                                   _lib.ptr(rays), int(f64), _lib.stream_ptr(device)), "mi_gen_rays")
    return rays


def sample_coarse(n: int, near: float, far: float, n_coarse: int, device, t_rand=None, seed: int = 0,
                  exact_linspace: bool = True, ray0: int = 0) -> torch.Tensor:
    """Stratified depths z[n,Nc] (render.py:123-132).  t_rand[n,Nc] injects the jitter; otherwise an
    in-kernel Philox stream keyed by (``seed``, ``ray0`` + ray index, sample) is used, ``ray0`` being the index of
    the first ray in the caller's full ray list (so the jitter of a frame does not depend on how it is split)."""
    lib = _lib.load()
    z = torch.empty((n, n_coarse), dtype=torch.float32, device=device)
    if t_rand is not None:
        t_rand = _f32c(t_rand, device)
        if tuple(t_rand.shape) != (n, n_coarse):
            raise _lib.MiRenderError(f"t_rand must be [{n},{n_coarse}]")
    zl = linspace_table(near, far, n_coarse, device) if exact_linspace else None
    with torch.cuda.device(device):
        _lib.check(lib.mi_sample_coarse(n, float(near), float(far), n_coarse, _lib.ptr(zl), _lib.ptr(t_rand),
                                        int(seed) & (2 ** 64 - 1), int(ray0), _lib.ptr(z), _lib.stream_ptr(device)),
                   "mi_sample_coarse")
    return z


def composite(raw: torch.Tensor, z: torch.Tensor, rays: torch.Tensor, want_weights: bool = True):
    """raw_to_outputs (render.py:78-103): returns rgb[n,3], depth[n], acc[n], weights[n,S]|None."""
    lib = _lib.load()
    dev = raw.device
    n, s = z.shape
    raw, z, rays = _f32c(raw, dev), _f32c(z, dev), _f32c(rays, dev)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    w = torch.empty((n, s), dtype=torch.float32, device=dev) if want_weights else None
    with torch.cuda.device(dev):
        _lib.check(lib.mi_composite(n, s, _lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays), _lib.ptr(rgb), _lib.ptr(depth),
                                    _lib.ptr(acc), _lib.ptr(w), _lib.stream_ptr(dev)), "mi_composite")
    return rgb, depth, acc, w


def sample_fine(z_coarse: torch.Tensor, weights: torch.Tensor, near: float, far: float, n_fine: int,
                want_samples: bool = False, exact_linspace: bool = True):
    """sample_pdf(mids, weights[...,1:-1], Nf) + detach + sort(cat) (render.py:140-142).
    Returns z_fine[n,Nc+Nf] (and z_samples[n,Nf] when asked)."""
    lib = _lib.load()
    dev = z_coarse.device
    n, nc = z_coarse.shape
    z_coarse, weights = _f32c(z_coarse, dev), _f32c(weights, dev)
    z_fine = torch.empty((n, nc + n_fine), dtype=torch.float32, device=dev)
    zs = torch.empty((n, n_fine), dtype=torch.float32, device=dev) if want_samples else None
    zl = linspace_table(near, far, nc, dev) if exact_linspace else None
    ul = linspace_table(0.0, 1.0, n_fine, dev) if exact_linspace else None
    with torch.cuda.device(dev):
        _lib.check(lib.mi_sample_fine(n, float(near), float(far), nc, n_fine, _lib.ptr(zl), _lib.ptr(ul),
                                      _lib.ptr(z_coarse), _lib.ptr(weights), _lib.ptr(zs), _lib.ptr(z_fine),
                                      _lib.stream_ptr(dev)), "mi_sample_fine")
    return (z_fine, zs) if want_samples else z_fine


def sample_fine_pos(z_coarse: torch.Tensor, weights: torch.Tensor, near: float, far: float, n_fine: int,
                    exact_linspace: bool = True):
    """sample_fine that also says where every input went: (z_fine [n,Nc+Nf], z_samples [n,Nf], pos [n,Nc+Nf] int32 with
    pos[e] = index in z_fine of z_coarse[e] (e < Nc) or z_samples[e - Nc]).  For one field shared by both passes."""
    lib = _lib.load()
    dev = z_coarse.device
    n, nc = z_coarse.shape
    z_coarse, weights = _f32c(z_coarse, dev), _f32c(weights, dev)
    z_fine = torch.empty((n, nc + n_fine), dtype=torch.float32, device=dev)
    zs = torch.empty((n, n_fine), dtype=torch.float32, device=dev)
    pos = torch.empty((n, nc + n_fine), dtype=torch.int32, device=dev)
    zl = linspace_table(near, far, nc, dev) if exact_linspace else None
    ul = linspace_table(0.0, 1.0, n_fine, dev) if exact_linspace else None
    with torch.cuda.device(dev):
        _lib.check(lib.mi_sample_fine_pos(n, float(near), float(far), nc, n_fine, _lib.ptr(zl), _lib.ptr(ul),
                                          _lib.ptr(z_coarse), _lib.ptr(weights), _lib.ptr(zs), _lib.ptr(z_fine), _lib.ptr(pos),
                                          _lib.stream_ptr(dev)), "mi_sample_fine_pos")
    return z_fine, zs, pos


def merge_raw(raw_c: torch.Tensor, raw_s: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """raw_fine [n,Nc+Nf,4] in sorted order from the coarse pass's raw [n,Nc,4] and the field at z_samples [n,Nf,4]."""
    lib = _lib.load()
    dev = raw_c.device
    n, nc = raw_c.shape[0], raw_c.shape[1]
    nf = pos.shape[1] - nc
    raw_f = torch.empty((n, nc + nf, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_merge_raw(n, nc, nf, _lib.ptr(raw_c), _lib.ptr(raw_s) if nf else None, _lib.ptr(pos),
                                    _lib.ptr(raw_f), _lib.stream_ptr(dev)), "mi_merge_raw")
    return raw_f


def split_grad(g_raw_f: torch.Tensor, pos: torch.Tensor, nc: int, g_raw_c: torch.Tensor | None = None):
    """The transpose of merge_raw: (g_raw_coarse [n,Nc,4] - added onto `g_raw_c` in place when given -, g_raw_samples
    [n,Nf,4])."""
    lib = _lib.load()
    dev = g_raw_f.device
    n, s = pos.shape
    nf = s - nc
    acc = g_raw_c is not None
    if g_raw_c is None:
        g_raw_c = torch.empty((n, nc, 4), dtype=torch.float32, device=dev)
    g_s = torch.empty((n, nf, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_split_grad(n, nc, nf, _lib.ptr(g_raw_f), _lib.ptr(pos), _lib.ptr(g_raw_c), int(acc),
                                     _lib.ptr(g_s) if nf else None, _lib.stream_ptr(dev)), "mi_split_grad")
    return g_raw_c, g_s


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_samples: int) -> torch.Tensor:
    """sample_pdf(bins[n,nb], weights[n,nb-1], N) -> [n,N] (render.py:27-56), any bins."""
    lib = _lib.load()
    dev = bins.device
    bins, weights = _f32c(bins, dev), _f32c(weights, dev)
    n, nb = bins.shape
    if tuple(weights.shape) != (n, nb - 1):
        raise _lib.MiRenderError("weights must be [n, len(bins)-1]")
    out = torch.empty((n, n_samples), dtype=torch.float32, device=dev)
    ul = linspace_table(0.0, 1.0, n_samples, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_sample_pdf(n, nb, n_samples, _lib.ptr(bins), _lib.ptr(weights), _lib.ptr(ul), _lib.ptr(out),
                                     _lib.stream_ptr(dev)), "mi_sample_pdf")
    return out


def _film_for(pf: PackedField, film, n_rays):
    if not is_film(pf.kind):
        return None, 1, n_rays
    if film is None:
        raise ValueError
    film = _f32c(film, pf.device).reshape(-1, 9, 512)
    groups = film.shape[0]
    if n_rays % groups:
        raise _lib.MiRenderError("rays must split evenly over the FiLM groups (images)")
    return film, groups, n_rays // groups


def field_eval_rays(pf: PackedField, rays: torch.Tensor, z: torch.Tensor, film=None) -> torch.Tensor:
    """run_network on points o + d*z with view dirs d/|d| (render.py:122,134-135), fused.  raw[n,S,4]."""
    lib = _lib.load()
    dev = pf.device
    rays, z = _f32c(rays, dev), _f32c(z, dev)
    n, s = z.shape
    film, groups, rpg = _film_for(pf, film, n)
    raw = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_field_eval_rays(pf.kind, _lib.ptr(pf.refresh()), _lib.ptr(film), _lib.ptr(rays), _lib.ptr(z),
                                          groups, rpg, s, _lib.ptr(raw), _lib.stream_ptr(dev)), "mi_field_eval_rays")
    return raw


class _Workspace:
    """Grow-only scratch for mi_render_rays (no allocation inside the library), one per (device, stream): launches
    on one stream are ordered, so they can share it; concurrent callers (DataParallel's thread per GPU,
    pi_GAN/train.py:50, or user threads on their own streams) never do."""
    bufs: dict = {}

    @classmethod
    def release(cls, device=None):
        """Drop the scratch of `device` (all devices if None): it only grows otherwise (3.4 GB after an 800x800 frame).
        Safe at any time between calls; the next render_rays allocates what it needs."""
        want = None if device is None else cls._name(device)
        for key in [k for k in cls.bufs if want is None or k[0] == want]:
            del cls.bufs[key]

    @staticmethod
    def _name(device) -> str:
        """'cuda', 'cuda:0', torch.device('cuda'), 0 -> 'cuda:<index>' (the current device where none is given)."""
        d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if d.type == "cuda" and d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        return str(d)

    @classmethod
    def get(cls, device, nbytes):
        key = (cls._name(device), torch.cuda.current_stream(device).cuda_stream)
        b = cls.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            cls.bufs[key] = b
        return b


def render_rays_fused(pf_c: PackedField, pf_f: PackedField, rays: torch.Tensor, near: float, far: float,
                      n_coarse: int, n_fine: int, film=None, t_rand=None, seed: int = 0, exact_linspace: bool = True,
                      ray0: int = 0):
    """render_rays (render.py:106-147) as one C-ABI call: six launches on the current stream.
    Returns the reference's 6-tuple."""
    lib = _lib.load()
    dev = pf_c.device
    rays = _f32c(rays, dev).reshape(-1, 2, 3)
    n = rays.shape[0]
    film, groups, rpg = _film_for(pf_c, film, n) if is_film(pf_c.kind) else _film_for(pf_f, film, n)
    if t_rand is not None:
        t_rand = _f32c(t_rand, dev)
        if tuple(t_rand.shape) != (n, n_coarse):
            raise _lib.MiRenderError(f"t_rand must be [{n},{n_coarse}]")
    outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in ((n, 3), (n,), (n,), (n, 3), (n,), (n,))]
    ws_bytes = lib.mi_render_workspace_bytes(n, n_coarse, n_fine)
    if pf_c is pf_f and n_fine > 0:
        # one field for both passes: with room for z_samples / their raw values / the merge positions the library evaluates
        # the Nf new depths only (mi_render_rays, include/mi_render.h); without it the plain path, same results
        ws_bytes += lib.mi_render_shared_field_extra_bytes(n, n_coarse, n_fine)
    guard = None
    if os.environ.get("MI_DEBUG_GUARDS") == "1":       # sentinel zone behind the workspace, checked after the call
        ws = torch.empty(int(ws_bytes) + 16384, dtype=torch.uint8, device=dev)
        ws[int(ws_bytes):] = 0xA5
        guard = ws[int(ws_bytes):]
    else:
        ws = _Workspace.get(dev, ws_bytes)
    zl = linspace_table(near, far, n_coarse, dev) if exact_linspace else None
    ul = linspace_table(0.0, 1.0, n_fine, dev) if exact_linspace else None
    with torch.cuda.device(dev):
        _lib.check(lib.mi_render_rays(pf_c.kind, _lib.ptr(pf_c.refresh()), pf_f.kind, _lib.ptr(pf_f.refresh()),
                                      _lib.ptr(film), _lib.ptr(rays), groups, rpg, float(near), float(far),
                                      n_coarse, n_fine, _lib.ptr(zl), _lib.ptr(ul), _lib.ptr(t_rand),
                                      int(seed) & (2 ** 64 - 1), int(ray0), *[_lib.ptr(o) for o in outs], _lib.ptr(ws),
                                      int(ws_bytes), _lib.stream_ptr(dev)), "mi_render_rays")
    if guard is not None and not bool((guard == 0xA5).all()):
        raise _lib.MiRenderError("mi_render_rays wrote past mi_render_workspace_bytes")
    return tuple(outs)
