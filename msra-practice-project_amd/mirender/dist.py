"""Multi-GPU: one process per GPU, rays sharded by contiguous ranges, RCCL all-gather of the frame.

Replaces the reference's only multi-device mechanism, torch.nn.DataParallel (pi_GAN/train.py:50,52:
per-step parameter broadcast + output gather on one process), for the render path: rays are independent
(SURVEY.md §8e), weights are replicated (<= 2.4 MB per model), each rank generates its own rays on the
device from (W, H, focal, c2w) so nothing is scattered, and ONE all-gather of packed [n_local, 5] fp32
(rgb, depth, acc; 1.6 MB per GPU for an 800x800 frame on 8 GPUs) reassembles the frame on every rank.
`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# With one rank there is nothing to exchange and the collectives are skipped - unless this switch is on (environment
# MI_FORCE_COLLECTIVE=1, `bench.py --force-collective`, tests/test_gpu_rccl.py): then a one-rank group still issues
# every collective, so the RCCL calls of this module execute (and are checked bit for bit against the ungrouped
# result) on a one-GPU box, before an 8-GPU node ever sees them.
FORCE_COLLECTIVE = os.environ.get("MI_FORCE_COLLECTIVE") == "1"


def _exchange(group=None) -> bool:
    """True when a collective has to be issued: a group exists and it has peers (or FORCE_COLLECTIVE is on)."""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVE)


def shard_range(total: int, rank: int, world: int):
    """Contiguous [start, stop) of `total` rays for `rank`; the first total % world ranks get one more."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rays(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """local [n_local, C] (this rank's shard_range rows) -> [total, C] on every rank, one collective.
    Shards are padded to the largest shard so a single all_gather_into_tensor moves everything."""
    if not _exchange(group):
        return local
    world = dist.get_world_size(group)
    width = local.shape[1]
    longest = -(-total // world)
    send = local
    if local.shape[0] != longest:
        send = torch.zeros((longest, width), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = torch.empty((world * longest, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if total == world * longest:
        return recv
    parts = []
    for r in range(world):
        a, b = shard_range(total, r, world)
        parts.append(recv[r * longest:r * longest + (b - a)])
    return torch.cat(parts)


def render_image_sharded(render_shard, width: int, height: int, group=None, timing: list | None = None):
    """Render this rank's ray range with `render_shard(ray0, n) -> (rgb[n,3], depth[n], acc[n])` and
    all-gather the frame.  Returns device tensors rgb[H,W,3], depth[H,W,1], acc[H,W,1] on every rank.
    `timing`: a list that receives one (start, end) pair of torch.cuda.Event per call, recorded on the current
    stream around the collective (bench.py reports the all-gather's cost from them)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    total = width * height
    a, b = shard_range(total, rank, world)
    rgb, depth, acc = render_shard(a, b - a)
    packed = torch.cat([rgb.reshape(-1, 3), depth.reshape(-1, 1), acc.reshape(-1, 1)], 1)
    if _exchange(group) and timing is not None and packed.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        full = all_gather_rays(packed, total, group)
        ev[1].record()
        timing.append(ev)
    else:
        full = all_gather_rays(packed, total, group) if _exchange(group) else packed
    return (full[:, :3].reshape(height, width, 3), full[:, 3:4].reshape(height, width, 1),
            full[:, 4:5].reshape(height, width, 1))


def render_image_dist(width, height, focal, pose, near, far, coarse_model, fine_model, n_coarse, n_fine,
                      seed=0, t_rand=None, group=None, timing: list | None = None):
    """nerf/render.py:150-167 sharded over the process group (fine-pass outputs, device tensors)."""
    from . import render_core

    def shard(ray0, n):
        tr = None if t_rand is None else t_rand[ray0:ray0 + n]
        with torch.no_grad():
            return render_core._render_image_device(width, height, focal, pose, near, far, coarse_model, fine_model,
                                                    n_coarse, n_fine, None, tr, seed, ray0, n)
    return render_image_sharded(shard, width, height, group, timing)


def allreduce_grads(params, group=None, timing: list | None = None):
    """Average renderer gradients over ranks in ONE flat all-reduce (4.75 MB for two NeRFs, 8.4 MB for the
    pi_GAN generator: latency-bound on xGMI, so one bucket).
    Every parameter that requires grad takes part, a missing gradient as zeros (and receives the average): the flat
    buffer then has the same layout on every rank whatever each rank's shard happened to touch - a rank whose shard is
    empty, or whose loss did not reach one of the models, would otherwise reduce a shorter buffer and hang the others.
    `timing`: a list that receives one (start, end) pair of torch.cuda.Event per call, recorded on the current stream
    around flatten + all-reduce + scatter-back (bench.py reports the gradient exchange's cost from them)."""
    if not _exchange(group):
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    ev = None
    if timing is not None and params[0].is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    flat = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    views, off = [], 0
    for p in params:
        views.append(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    for p, g in zip(params, views):
        if p.grad is None:
            p.grad = torch.empty_like(p)
    # one multi-tensor copy instead of one small kernel per parameter (48 for two NeRFs: 0.1 ms of a 7 ms step otherwise)
    torch._foreach_copy_([p.grad for p in params], views)
    if ev is not None:
        ev[1].record()
        timing.append(ev)
    return flat.numel() * flat.element_size()
