"""Oracle (test infrastructure): synthetic weights, poses and jitter.

No dataset or checkpoint ships with the reference, so every
parity / bench input is synthetic.  Weights follow the reference initialisers'
*distributions* (cited per branch) but are drawn from NumPy PCG64 so the same
seed gives the same bytes here, in make_golden.py (where they are loaded into
the imported reference modules) and on the GPU box.
"""
from __future__ import annotations

import hashlib

import numpy as np
import torch

from .fields import SPECS


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _layer_init(kind: str, key: str, o: int, i: int, rng):
    """Return (weight, bias) float32 arrays for one linear layer."""
    if kind in ("nerf", "tiny_nerf"):
        # Dense.reset_parameters, nerf/nerf.py:25-28: xavier_uniform(gain(act)), zero bias
        act = "relu"
        if key == "output_layer_rgb":
            act = "sigmoid"
        if kind == "nerf" and key == "layers_dir.0":
            act = "linear"
        gain = np.sqrt(2.0) if act == "relu" else 1.0
        return _uniform(rng, (o, i), gain * np.sqrt(6.0 / (i + o))), np.zeros(o, np.float32)
    if kind == "siren_nerf":
        if key in ("layers_dir.0", "output_layer_sigma", "output_layer_rgb"):
            gain = np.sqrt(2.0) if key == "output_layer_sigma" else 1.0
            return _uniform(rng, (o, i), gain * np.sqrt(6.0 / (i + o))), np.zeros(o, np.float32)
        # Siren.reset_parameters nerf/nerf.py:114-117; first layer override nerf/nerf.py:134
        bound = 1.0 / 30.0 if key == "layers_pos.0" else np.sqrt(6.0 / i) / 30.0
        return _uniform(rng, (o, i), bound), np.zeros(o, np.float32)
    if kind.startswith("film_siren_nerf"):
        if key in ("output_layer_sigma.0", "output_layer_rgb.0"):
            # torch.nn.Linear default: U(+-1/sqrt(in)) for weight and bias
            b = 1.0 / np.sqrt(i)
            return _uniform(rng, (o, i), b), _uniform(rng, (o,), b)
        # FilmSiren.reset_parameters pi_GAN/modules.py:27-31
        wb = 1.0 / i if key == "input_layer" else np.sqrt(6.0 / i) / 30.0
        return _uniform(rng, (o, i), wb), _uniform(rng, (o,), np.sqrt(1.0 / i))
    raise KeyError(kind)


SIGMA_HEAD = {True: (50.0, 5.0), "medium": (8.0, 2.0)}   # sharp -> (weight scale, bias shift) of the sigma head


def state_dict(kind: str, seed: int = 0, sharp=False, bias_jitter: float = 0.0) -> dict:
    """Synthetic fp32 state dict (torch CPU tensors) with the reference's key layout.

    sharp: True scales the sigma head x50 and adds +5 bias so the volume is not near-empty
    (SURVEY.md §8d); "medium" (x8, +2) gives a volume that turns opaque within ~15 samples while the fp32
    pipeline stays well inside 1e-4 of exact arithmetic (the x50 head amplifies hidden-layer rounding into
    2e-4 of alpha).  bias_jitter: add U(+-bias_jitter) to every bias so bias paths
    are exercised (the reference initialises Dense/Siren biases to zero)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, (o, i) in SPECS[kind]:
        w, b = _layer_init(kind, key, o, i, rng)
        if bias_jitter:
            b = b + _uniform(rng, (o,), bias_jitter)
        if sharp and key.startswith("output_layer_sigma"):
            scale, shift = SIGMA_HEAD[sharp]
            w = w * scale
            b = b + shift
        sd[key + ".weight"] = torch.from_numpy(np.ascontiguousarray(w))
        sd[key + ".bias"] = torch.from_numpy(np.ascontiguousarray(b.astype(np.float32)))
    return sd


def film_params(n_images: int, seed: int = 1, spread: float = 0.25) -> torch.Tensor:
    """[n_images, 9, 512] FiLM table shaped like MappingNetwork output
    (pi_GAN/modules.py:61-68); gamma around 1, beta around 0 as its bias init (:56-58)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f = _uniform(rng, (n_images, 9, 512), spread)
    f[:, :, :256] += 1.0
    return torch.from_numpy(f)


def t_rand(n_rays: int, n_coarse: int, seed: int = 123) -> torch.Tensor:
    """Stratified jitter U[0,1) fp32, injected into both oracle and HIP path."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random((n_rays, n_coarse), dtype=np.float32))


def digest(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
    return h.hexdigest()


# -- camera helpers -----------------------------------------------------------
def _trans_z(t):
    m = np.eye(4, dtype=np.float32)
    m[2, 3] = t
    return m


def _pitch(phi):
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def _yaw(th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def pose_radians(radius, theta, phi):
    """pi_GAN/render.py:37-49 (angles in radians)."""
    return _yaw(theta) @ (_pitch(phi) @ _trans_z(radius))


def pose_degrees(radius, theta, phi):
    """nerf/data_loader.py:39-51 (angles in degrees)."""
    return pose_radians(radius, theta / 180.0 * np.pi, phi / 180.0 * np.pi)
