"""Oracle (test infrastructure): a teacher-field scene and the reference's training loop on the CPU.

No dataset ships with the reference, so the "PSNR within 0.05 dB of the reference" clause of the
north star is checked on a synthetic scene: images of a fixed seeded TEACHER field rendered by the oracle, a
student fitted to them with the loop of nerf/train_nerf.py:124-176 (ray batches, render_rays, MSE fine + coarse,
Adam with the decayed learning rate), once by the caller's renderer and once here by CPU autograd through the
oracle, same initialisation, same batches, same injected jitter.  Only tests/ and bench.py's reporting import this.
"""
from __future__ import annotations

import numpy as np
import torch

from . import fields as ofields, render_ref as R, synth

NEAR, FAR = 2.0, 6.0


class Scene:
    """`n_views` training views + one held-out view of a teacher tiny_nerf ("medium" density), res x res pixels;
    the student pair is of kind `student` (tiny_nerf: PE + ReLU, or siren_nerf: the sin(30 u) family)."""

    def __init__(self, res=24, n_views=6, nc=16, nf=16, seed=100, student="tiny_nerf", images=None):
        """`images` [n_views + 1, res*res, 3]: the teacher's views as rendered elsewhere (fixture fit_r03_scene.npz: the
        build container's render).  The teacher is rendered by the oracle through the ill-conditioned hierarchical
        resampling, so two hosts' CPUs paint a few rays differently (1e-4 of the loss): a trajectory recorded on one host
        can only be compared with a run that fits the SAME pictures."""
        self.res, self.nc, self.nf, self.student = res, nc, nf, student
        self.focal = 1.3875 * res
        # FiLM students: one fixed FiLM row [9,512] (a single "image" of the mapping network's output, gamma ~ 1,
        # beta ~ 0: pi_GAN/modules.py:56-58) shared by every ray - the field alone is trained, as synthesis.py:83-107
        # trains through a fixed generator
        self.film = synth.film_params(1, seed=seed + 3)[0] if student.startswith("film") else None
        sd_t = synth.state_dict("tiny_nerf", seed=seed, sharp="medium", bias_jitter=0.05)
        teacher = ofields.make_field("tiny_nerf", sd_t)
        angles = list(np.linspace(-150.0, 150.0, n_views)) + [17.0]
        self.poses = [synth.pose_degrees(4.0, float(a), -30.0) for a in angles]
        self.rays = [torch.from_numpy(R.rays_from_camera(res, res, self.focal, p)) for p in self.poses]
        imgs = []
        if images is not None:
            imgs = [torch.as_tensor(np.asarray(im), dtype=torch.float32) for im in images]
            assert len(imgs) == n_views + 1 and tuple(imgs[0].shape) == (res * res, 3)
        with torch.no_grad():
            for i, r in enumerate(self.rays if images is None else []):
                t = R.render_rays(r, NEAR, FAR, teacher, teacher, 32, 64, synth.t_rand(res * res, 32, seed=7000 + i))
                imgs.append(t.rgb_f)
        self.images = imgs                                    # [res*res, 3] each; the last one is held out
        self.train_rays = torch.cat(self.rays[:-1])
        self.train_rgb = torch.cat(self.images[:-1])
        self.student_init = (synth.state_dict(student, seed=seed + 1, bias_jitter=0.02),
                             synth.state_dict(student, seed=seed + 2, bias_jitter=0.02))

    def batch(self, step: int, batch_size: int):
        """Deterministic batches: a fixed permutation of all training rays, walked in order (train_nerf.py:140-147)."""
        n = self.train_rays.shape[0]
        if not batch_size:
            return self.train_rays, self.train_rgb, synth.t_rand(n, self.nc, seed=90000 + step)
        perm = np.random.Generator(np.random.PCG64(4242)).permutation(n)
        idx = torch.from_numpy(perm[(step * batch_size) % n:][:batch_size].copy())
        if idx.numel() < batch_size:
            idx = torch.from_numpy(perm[:batch_size].copy())
        return self.train_rays[idx], self.train_rgb[idx], synth.t_rand(batch_size, self.nc, seed=90000 + step)

    def heldout_jitter(self):
        return synth.t_rand(self.res * self.res, self.nc, seed=555)


def lr_at(step, lr0=5e-4, decay=250):
    return lr0 * (0.1 ** (step / (decay * 1000)))            # train_nerf.py:170-173


def make_optimizer(params, optimizer: str, lr0: float):
    """"adam": torch.optim.Adam as train_nerf.py:98 builds it.  "sgd": plain gradient descent - Adam's first updates are
    m/sqrt(v) = +-1 per element whatever the gradient's size, so only a loop whose step is PROPORTIONAL to the gradient
    shows a gradient of the wrong magnitude in the next step's loss."""
    if optimizer == "sgd":
        return torch.optim.SGD(params, lr=lr0)
    return torch.optim.Adam(params, lr=lr0, betas=(0.9, 0.999))


def fit_cpu(scene: Scene, steps: int, batch_size: int, f64: bool = False, lr0: float = 5e-4, optimizer: str = "adam"):
    """The reference loop on CPU autograd through the oracle.  Returns (losses[steps], heldout_psnr, state dicts).
    f64: the same loop with weights, activations and optimiser state in double - how far apart two correct
    implementations of this loop may drift (sin networks amplify rounding differences from the first Adam steps on).
    batch_size 0: every training ray in every step (no batching noise: the quietest regime of the loop)."""
    cast = (lambda v: v.double()) if f64 else (lambda v: v)
    sd_c = {k: cast(v).clone().requires_grad_(True) for k, v in scene.student_init[0].items()}
    sd_f = {k: cast(v).clone().requires_grad_(True) for k, v in scene.student_init[1].items()}
    film = None if scene.film is None else cast(scene.film)
    fc, ff = ofields.make_field(scene.student, sd_c, film), ofields.make_field(scene.student, sd_f, film)
    render = R.render_rays_f64 if f64 else R.render_rays
    params = list(sd_c.values()) + list(sd_f.values())
    opt = make_optimizer(params, optimizer, lr0)
    losses = []
    for step in range(steps):
        rays, rgb, tr = scene.batch(step, batch_size)
        out = render(rays, NEAR, FAR, fc, ff, scene.nc, scene.nf, tr)
        rgb = cast(rgb)
        loss = torch.mean((out.rgb_f - rgb) ** 2) + torch.mean((out.rgb_c - rgb) ** 2)      # :158-166
        opt.zero_grad()
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g["lr"] = lr_at(step + 1, lr0)
        losses.append(float(loss.detach()))
    with torch.no_grad():
        held = render(scene.rays[-1], NEAR, FAR, fc, ff, scene.nc, scene.nf, scene.heldout_jitter())
    psnr = R.psnr(held.rgb_f.numpy(), scene.images[-1].numpy())
    return losses, psnr, ({k: v.detach() for k, v in sd_c.items()}, {k: v.detach() for k, v in sd_f.items()})
