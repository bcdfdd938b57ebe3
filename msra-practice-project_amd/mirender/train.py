"""The data path of nerf/train_nerf.py's loop around render_rays, on the device:

  RayBank      the rays_rgba table and its batches         train_nerf.py:78-86, 143-147
  nerf_loss    loss + psnr of one batch, one kernel        train_nerf.py:158-167
  decayed_lr   the learning-rate schedule                  train_nerf.py:170-175

The optimiser stays torch.optim.Adam (train_nerf.py:96); pass fused=True to update all tensors in one launch.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class RayBank:
    """rays_rgba [N*H*W, 10] = (rays_o, rays_d, r, g, b, a) for every pixel of every training image, built and
    shuffled on the device.  images [N,H,W,4] (RGBA in [0,1]), poses [N,>=3,4] camera-to-world, focal a float.

    Unlike the reference the per-epoch reshuffle takes effect (train_nerf.py:144 assigns the permuted table to
    a misspelt name, so every epoch replays the first one)."""

    def __init__(self, images, poses, focal: float, device="cuda", white_bkgd: bool = True, generator=None):
        lib = _lib.load()
        dev = torch.device(device)
        images = torch.as_tensor(np.asarray(images) if not isinstance(images, torch.Tensor) else images)
        images = images.to(device=dev, dtype=torch.float32).contiguous()
        n, h, w, c = images.shape
        if c != 4:
            raise _lib.MiRenderError("RayBank expects RGBA images [N,H,W,4]")
        poses = torch.as_tensor(np.asarray(poses) if not isinstance(poses, torch.Tensor) else poses)
        poses = poses.to(dtype=torch.float32)[:, :3, :4].contiguous().reshape(n, 12).to(dev)
        self.width, self.height, self.generator = w, h, generator
        self.table = torch.empty((n * h * w, 10), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mi_ray_bank(w, h, float(focal), _lib.ptr(poses), _lib.ptr(images), int(white_bkgd), n,
                                       _lib.ptr(self.table), _lib.stream_ptr(dev)), "mi_ray_bank")
        self.batch_idx = 0

    def __len__(self):
        return self.table.shape[0]

    def shuffle(self):
        """np.random.shuffle(rays_rgba) (train_nerf.py:83) / the epoch reshuffle (:143-145), on the device."""
        perm = torch.randperm(len(self), device=self.table.device, generator=self.generator)
        self.table = self.table[perm]
        self.batch_idx = 0

    def batch(self, batch_size: int):
        """Next (batch_rays [B,2,3], batch_rgb [B,3], batch_alpha [B]) as train_nerf.py:140-150 slices them;
        reshuffles when the table is exhausted."""
        n_batches = -(-len(self) // batch_size)
        b = self.table[self.batch_idx * batch_size:(self.batch_idx + 1) * batch_size]
        self.batch_idx += 1
        if self.batch_idx == n_batches:
            self.shuffle()
        return b[:, :6].reshape(-1, 2, 3), b[:, 6:9], b[:, 9]


class _NerfLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_c, acc_c, rgb_f, acc_f, target, use_alpha, use_fine):
        lib = _lib.load()
        dev = rgb_f.device
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        rgb_c, acc_c, rgb_f, acc_f, target = f(rgb_c), f(acc_c), f(rgb_f), f(acc_f), f(target)
        n = rgb_f.shape[0]
        grads = [torch.empty_like(t) for t in (rgb_c, acc_c, rgb_f, acc_f)]
        ws = torch.empty(lib.mi_nerf_loss_workspace_floats(n), dtype=torch.float32, device=dev)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mi_nerf_loss(n, _lib.ptr(rgb_c), _lib.ptr(acc_c), _lib.ptr(rgb_f), _lib.ptr(acc_f),
                                        _lib.ptr(target), int(use_alpha), int(use_fine), *[_lib.ptr(g) for g in grads],
                                        _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr(dev)), "mi_nerf_loss")
        ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        g = [t * g_loss for t in ctx.saved_tensors]
        return g[0], g[1], g[2], g[3], None, None, None


def nerf_loss(outputs, batch_rgb, batch_alpha, use_alpha: bool = False, use_fine_model: bool = True):
    """train_nerf.py:158-167 on render_rays' 6-tuple.  Returns (loss, psnr); loss.backward() seeds render_rays'
    backward with gradients computed in the same kernel."""
    rgb_c, _, acc_c, rgb_f, _, acc_f = outputs
    if not rgb_f.is_cuda:
        raise _lib.MiRenderError("nerf_loss needs the render outputs on a ROCm device (there is no CPU path)")
    target = torch.cat([batch_rgb.reshape(-1, 3), batch_alpha.reshape(-1, 1)], 1)
    loss, out = _NerfLossFn.apply(rgb_c, acc_c, rgb_f, acc_f, target, bool(use_alpha), bool(use_fine_model))
    return loss, -10.0 * torch.log10(out[1])


def decayed_lr(learning_rate: float, learning_rate_decay: float, global_step: int, decay_rate: float = 0.1) -> float:
    """train_nerf.py:170-173: lr * 0.1 ** (step / (learning_rate_decay * 1000))."""
    return learning_rate * (decay_rate ** (global_step / (learning_rate_decay * 1000)))
