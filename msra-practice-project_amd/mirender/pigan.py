"""pi-GAN generator side on the HIP renderer: Renderer / Generator of pi_GAN/modules.py:121-197.

Same constructor arguments, attributes and call surface as the reference classes.  Two deliberate
differences in HOW the work is issued (results are the same function of the inputs):

* `Generator.forward` renders the whole batch in ONE launch sequence - rays of all b images concatenated,
  per-image FiLM tables consumed directly by the fused kernel as b groups - instead of
  the reference's sequential per-image loop with module state (modules.py:179-181);
* camera angles are still drawn from NumPy's global RNG, in the reference's order (theta then phi per
  image, modules.py:155-158), so `np.random.seed` reproduces the same poses.

The mapping network (modules.py:34-68) is a plain 3-layer MLP with nine 512-wide heads; it stays on
PyTorch-ROCm (three tiny GEMMs per batch) and is outside the hand-written path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import fields, ops, render_core


def camera_pos_to_transform_matrix(radius, theta, phi):
    """pi_GAN/render.py:37-49 (radians): yaw(theta) @ pitch(phi) @ translate_z(radius), float32."""
    t = np.eye(4, dtype=np.float32)
    t[2, 3] = radius
    c, s = np.cos(phi), np.sin(phi)
    rp = np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float32)
    c, s = np.cos(theta), np.sin(theta)
    rt = np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float32)
    return rt @ (rp @ t)


def render_batch(model, film, poses, width, height, focal, near, far, n_coarse, n_fine, t_rand=None, seed=None):
    """b images [b,H,W,3] (fine rgb, autograd graph attached) from one FiLM field, film [b,9,512], b poses."""
    pf = fields.as_packed_field(model)
    dev = pf.device
    b = len(poses)
    rays = torch.cat([ops.gen_rays(width, height, focal, p, dev) for p in poses])
    out = render_core.render_rays(rays, near, far, model, model, n_coarse, n_fine, t_rand=t_rand, seed=seed,
                                  film=film.reshape(b, 9, 512))
    return out[3].reshape(b, height, width, 3)


class Renderer:
    """pi_GAN/modules.py:121-162."""

    def __init__(self, width, height, near=0.1, far=1.9, fov=12, coarse_samples=64, fine_samples=128,
                 horizontal_std=0.3, vertical_std=0.15):
        self.width, self.height, self.fov = width, height, fov
        self.focal = width / 2 / np.tan(fov / 2 * np.pi / 180)        # np.float64 scalar, as in the reference
        self.near, self.far = near, far
        self.coarse_samples, self.fine_samples = coarse_samples, fine_samples
        self.horizontal_std, self.vertical_std = horizontal_std, vertical_std

    def set_params(self, width=None, height=None, near=None, far=None, fov=None, coarse_samples=None,
                   fine_samples=None, horizontal_std=None, vertical_std=None):
        for k, v in dict(width=width, height=height, near=near, far=far, fov=fov, coarse_samples=coarse_samples,
                         fine_samples=fine_samples, horizontal_std=horizontal_std, vertical_std=vertical_std).items():
            if v is not None:
                setattr(self, k, v)
        if width is not None or fov is not None:
            self.focal = self.width / 2 / np.tan(self.fov / 2 * np.pi / 180)

    def sample_pose(self, theta=None, phi=None):
        if theta is None:
            theta = np.random.randn() * self.horizontal_std
        if phi is None:
            phi = np.random.randn() * self.vertical_std
        return camera_pos_to_transform_matrix(1, theta, phi)

    def __call__(self, model, theta=None, phi=None):
        pose = self.sample_pose(theta, phi)
        return render_core.render_image_tensor(self.width, self.height, self.focal, pose, self.near, self.far, model,
                                               model, self.coarse_samples, self.fine_samples)

    def render_batch(self, model, film, thetas=None, phis=None, t_rand=None, seed=None):
        b = film.shape[0]
        poses = [self.sample_pose(None if thetas is None else thetas[i], None if phis is None else phis[i])
                 for i in range(b)]
        return render_batch(model, film, poses, self.width, self.height, self.focal, self.near, self.far,
                            self.coarse_samples, self.fine_samples, t_rand, seed)


class MappingNetwork(torch.nn.Module):
    """pi_GAN/modules.py:34-68: z -> 9 x (gamma | beta); state-dict keys as the reference's."""

    def __init__(self, input_dim=256, output_dim=256, output_layers=8, hidden_dim=256, hidden_layers=3):
        super().__init__()
        act = lambda: torch.nn.LeakyReLU(0.2)  # noqa: E731
        self.input_layer = torch.nn.Sequential(torch.nn.Linear(input_dim, hidden_dim), act())
        hid = []
        for _ in range(hidden_layers - 1):
            hid += [torch.nn.Linear(hidden_dim, hidden_dim), act()]
        self.hidden_layers = torch.nn.Sequential(*hid)
        heads = [torch.nn.Linear(hidden_dim, 2 * output_dim) for _ in range(output_layers + 1)]
        for h in heads:                       # modules.py:56-58: gamma starts at 1, beta at 0
            h.bias.data[:output_dim] = 1
            h.bias.data[output_dim:] = 0
        self.output_layers = torch.nn.ModuleList(heads)

    def forward(self, z):
        h = self.hidden_layers(self.input_layer(z))
        return torch.stack([head(h) for head in self.output_layers], dim=1)


class Generator(torch.nn.Module):
    """pi_GAN/modules.py:165-197 with the batch rendered in one launch sequence."""

    def __init__(self, input_dim, output_size, near=0.1, far=1.9, fov=12, coarse_samples=64, fine_samples=128,
                 horizontal_std=0.3, vertical_std=0.15, use_dir=True):
        super().__init__()
        self.input_dim = input_dim
        self.film_siren_nerf = fields.FilmSirenNeRF(use_dir=use_dir)
        self.mapping_network = MappingNetwork(input_dim=input_dim)
        self.renderer = Renderer(output_size, output_size, near, far, fov, coarse_samples, fine_samples,
                                 horizontal_std, vertical_std)

    def forward(self, input_tensor, thetas=None, phis=None, t_rand=None, seed=None):
        film = self.mapping_network(input_tensor)                               # [b,9,512]
        img = self.renderer.render_batch(self.film_siren_nerf, film, thetas, phis, t_rand, seed)
        return img.permute(0, 3, 1, 2).contiguous()                             # [b,3,H,W] (modules.py:182-183)

    def get_mapping(self, input_tensor):
        return self.mapping_network(input_tensor)

    def set_film_params(self, film_params):
        self.film_siren_nerf.set_film_params(film_params)

    def set_resolution(self, resolution):
        self.renderer.set_params(width=resolution, height=resolution)

    def render(self, theta=None, phi=None):
        return self.renderer(self.film_siren_nerf, theta, phi)
