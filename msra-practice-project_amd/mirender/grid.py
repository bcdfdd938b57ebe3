"""Dense field queries on a voxel grid: the sampling half of create_mesh (pi_GAN/utils.py:42-96), which feeds
marching cubes with -sigma on an N^3 grid.  Points are generated on the device (mi_grid_points) and evaluated
by the fused field kernel in batches of `max_batch` (the reference's 64^3 default), nothing visits the host."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib, fields


def grid_points(n: int, voxel_origin, voxel_size: float, head: int, count: int, device) -> torch.Tensor:
    """Rows [head, head+count) of create_mesh's `samples` table (utils.py:57-71) + a zero view direction: [count,6]."""
    lib = _lib.load()
    origin = np.ascontiguousarray(np.asarray(voxel_origin, dtype=np.float32).reshape(3))
    pts = torch.empty((count, 6), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.mi_grid_points(int(n), origin.ctypes.data_as(ctypes.c_void_p), float(voxel_size), int(head),
                                      int(count), _lib.ptr(pts), _lib.stream_ptr(device)), "mi_grid_points")
    return pts


def density_grid(decoder, n: int = 256, max_batch: int = 64 ** 3, voxel_origin=(-0.1, -0.1, -0.1),
                 voxel_size: float | None = None, film=None) -> torch.Tensor:
    """`sdf_values` of create_mesh (utils.py:73-96): -sigma of `decoder` on the N^3 grid, [N,N,N] on the device.
    decoder: a known field (FilmSirenNeRF with film params set or `film` given, NeRF, SirenNeRF, ...)."""
    pf = fields.as_packed_field(decoder)
    if voxel_size is None:
        voxel_size = 0.2 / (n - 1)                       # utils.py:55
    if film is None and fields.is_film(pf.kind):
        film = fields.film_table(decoder)
    total = n ** 3
    out = torch.empty(total, dtype=torch.float32, device=pf.device)
    head = 0
    while head < total:
        count = min(max_batch, total - head)
        pts = grid_points(n, voxel_origin, voxel_size, head, count, pf.device)
        out[head:head + count] = fields.eval_points(pf, pts, film)[:, 3]
        head += count
    return (-out).reshape(n, n, n)
