"""Checkpoint interop with the reference's training scripts.

The reference saves plain dicts with torch.save (nerf/train_nerf.py:181-189: global_step, coarse_model, fine_model |
None, optimizer; pi_GAN/train.py:162-172: global_step, loss_log, generator, discriminator, g_optimizer, d_optimizer,
the two networks saved from DataParallel's .module) and resumes from the lexicographically last file whose name
contains 'tar' (train_nerf.py:101-114, pi_GAN/train.py:62-76).  Here the same dicts are written and read; loading
uses weights_only=True, so nothing in a checkpoint file is executed, and the field state dicts go straight into the
fused modules (fields.field_from_state_dict picks the class from the key/shape layout; its packed MFMA stream is
built on first use)."""
from __future__ import annotations

import os

import torch

from . import fields, pigan


def latest(log_path: str):
    """The file train_nerf.py:101-104 / pi_GAN/train.py:64-67 would resume from, or None."""
    names = [f for f in sorted(os.listdir(log_path)) if "tar" in f]
    return os.path.join(log_path, names[-1]) if names else None


def checkpoint_path(log_path: str, global_step: int) -> str:
    return os.path.join(log_path, "{:06d}.tar".format(global_step))          # train_nerf.py:182


def save_nerf(path: str, global_step: int, coarse_model, fine_model, optimizer) -> None:
    """The dict of nerf/train_nerf.py:183-188 (fine_model None when use_fine_model is off)."""
    torch.save({"global_step": global_step, "coarse_model": coarse_model.state_dict(),
                "fine_model": None if fine_model is None or fine_model is coarse_model else fine_model.state_dict(),
                "optimizer": optimizer.state_dict()}, path)


def save_pigan(path: str, global_step: int, loss_log: dict, generator, discriminator, g_optimizer, d_optimizer) -> None:
    """The dict of pi_GAN/train.py:164-171; `discriminator` may be None (it is outside the render path)."""
    unwrap = lambda m: getattr(m, "module", m)  # noqa: E731  (DataParallel / DistributedDataParallel)
    torch.save({"global_step": global_step, "loss_log": loss_log, "generator": unwrap(generator).state_dict(),
                "discriminator": None if discriminator is None else unwrap(discriminator).state_dict(),
                "g_optimizer": g_optimizer.state_dict(),
                "d_optimizer": None if d_optimizer is None else d_optimizer.state_dict()}, path)


def load(path: str) -> dict:
    """torch.load(weights_only=True) onto the CPU: tensors, numbers, lists and dicts only."""
    return torch.load(path, map_location="cpu", weights_only=True)


def nerf_models(check_point: dict, device="cuda"):
    """(coarse_model, fine_model) as fused modules from a train_nerf.py checkpoint dict; fine_model is the coarse one
    when the checkpoint holds None (use_fine_model off: train_nerf.py:91,94 aliases them)."""
    coarse = fields.field_from_state_dict(check_point["coarse_model"], device)
    fine_sd = check_point.get("fine_model")
    return coarse, coarse if fine_sd is None else fields.field_from_state_dict(fine_sd, device)


def pigan_generator(check_point: dict, output_size: int, device="cuda", **renderer_kw) -> "pigan.Generator":
    """Generator (pi_GAN/modules.py:165) rebuilt from a pi_GAN/train.py checkpoint dict: input_dim and use_dir come
    from the stored shapes; the renderer's settings are not part of a state dict and are passed by the caller."""
    sd = check_point["generator"]
    input_dim = sd["mapping_network.input_layer.0.weight"].shape[1]
    use_dir = sd["film_siren_nerf.hidden_layer_rgb.weight"].shape[1] == 259
    gen = pigan.Generator(input_dim, output_size, use_dir=use_dir, **renderer_kw)
    gen.load_state_dict(sd)
    return gen.to(device)
