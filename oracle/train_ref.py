"""CPU restatement of the data path of nerf/train_nerf.py's loop.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  rays_rgba   train_nerf.py:64-68 (white background), :78-82 (the [N*H*W, 10] table, before the shuffle)
  nerf_loss   train_nerf.py:158-167

The script itself cannot be imported (it hard-requires CUDA and a dataset at import: train_nerf.py:11,52), so
the loss is pinned by fixture F7, whose generator (tests/golden/make_golden.py) evaluates these very lines on
the reference's render_rays outputs; the table is get_rays (pinned by fixture F1) plus NumPy reshapes."""
import numpy as np
import torch

from . import render_ref


def rays_rgba(images, poses, width, height, focal, white_bkgd=True):
    images = np.array(images, dtype=np.float32, copy=True)
    if white_bkgd:
        images[..., :3] = images[..., :3] * images[..., -1:] + (1. - images[..., -1:])                # :68
    rays = np.stack([render_ref.get_rays(width, height, focal, p) for p in poses[:, :3, :4]], 0)      # :78
    rays = np.transpose(rays, [0, 2, 3, 1, 4])                                                        # :79
    rays = np.reshape(rays, [-1, 6])                                                                  # :80
    rgba = np.reshape(images, [-1, 4])                                                                # :81
    return np.concatenate([rays, rgba], 1).astype(np.float32)                                         # :82, :84


def nerf_loss(outputs, batch_rgb, batch_alpha, use_alpha=False, use_fine_model=True):
    rgb_map_coarse, _, acc_map_coarse, rgb_map_fine, _, acc_map_fine = outputs
    loss_coarse = torch.mean((rgb_map_coarse - batch_rgb) ** 2)                                       # :158
    loss_fine = torch.mean((rgb_map_fine - batch_rgb) ** 2)                                           # :159
    psnr = -10 * torch.log10(loss_fine)                                                               # :160
    if use_alpha:
        loss_coarse = loss_coarse + 0.1 * torch.mean((acc_map_coarse - batch_alpha) ** 2)             # :162
        loss_fine = loss_fine + 0.1 * torch.mean((acc_map_fine - batch_alpha) ** 2)                   # :163
    loss = loss_fine                                                                                  # :164
    if use_fine_model:
        loss = loss + loss_coarse                                                                     # :165-166
    return loss, psnr
