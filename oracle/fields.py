"""Oracle (test infrastructure): radiance-field MLPs restated functionally.

Each function maps points ``x[M,6] = (xyz, view_dir)`` to ``[M,4] = (r,g,b,sigma)``
given a plain ``dict[str, Tensor]`` whose keys/shapes equal the reference
modules' ``state_dict()`` layouts, so fixtures taken from the reference load
unchanged:

* ``nerf``            - reference nerf/nerf.py:52-94  (NeRF, PE(10)/PE(4), 8x256 ReLU, skip at 5)
* ``siren_nerf``      - reference nerf/nerf.py:120-170 (SirenNeRF, sin(30*.) layers, raw xyz/dir)
* ``film_siren_nerf`` - reference pi_GAN/modules.py:70-118 (FilmSirenNeRF) with the FiLM layer
                        of pi_GAN/modules.py:22-25: sin(w0*(gamma*(xW^T+b)+beta)), w0=30
* ``tiny_nerf``       - build-defined 4-layer variant for BASELINE config C1;
                        not present in the reference, rendered through the same contract.

Positional encoding follows nerf/nerf.py:44-49: concat over i<L of
[sin(2^i x), cos(2^i x)], no identity term.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

W0 = 30.0

# ----------------------------------------------------------------------------
# parameter layouts: (key, (out, in)) per linear layer, in forward order
# ----------------------------------------------------------------------------
SPECS = {
    "nerf": [
        ("layers_pos.0", (256, 60)), ("layers_pos.1", (256, 256)), ("layers_pos.2", (256, 256)),
        ("layers_pos.3", (256, 256)), ("layers_pos.4", (256, 256)), ("layers_pos.5", (256, 316)),
        ("layers_pos.6", (256, 256)), ("layers_pos.7", (256, 256)),
        ("layers_dir.0", (256, 256)), ("layers_dir.1", (128, 280)),
        ("output_layer_sigma", (1, 256)), ("output_layer_rgb", (3, 128)),
    ],
    "siren_nerf": [
        ("layers_pos.0", (256, 3)), ("layers_pos.1", (256, 256)), ("layers_pos.2", (256, 256)),
        ("layers_pos.3", (256, 256)), ("layers_pos.4", (256, 256)), ("layers_pos.5", (256, 259)),
        ("layers_pos.6", (256, 256)), ("layers_pos.7", (256, 256)),
        ("layers_dir.0", (256, 256)), ("layers_dir.1", (128, 259)),
        ("output_layer_sigma", (1, 256)), ("output_layer_rgb", (3, 128)),
    ],
    "film_siren_nerf": [
        ("input_layer", (256, 3)),
        ("hidden_layers.0", (256, 256)), ("hidden_layers.1", (256, 256)), ("hidden_layers.2", (256, 256)),
        ("hidden_layers.3", (256, 256)), ("hidden_layers.4", (256, 256)), ("hidden_layers.5", (256, 256)),
        ("hidden_layers.6", (256, 256)),
        ("output_layer_sigma.0", (1, 256)), ("hidden_layer_rgb", (256, 259)), ("output_layer_rgb.0", (3, 256)),
    ],
    "film_siren_nerf_nodir": [
        ("input_layer", (256, 3)),
        ("hidden_layers.0", (256, 256)), ("hidden_layers.1", (256, 256)), ("hidden_layers.2", (256, 256)),
        ("hidden_layers.3", (256, 256)), ("hidden_layers.4", (256, 256)), ("hidden_layers.5", (256, 256)),
        ("hidden_layers.6", (256, 256)),
        ("output_layer_sigma.0", (1, 256)), ("hidden_layer_rgb", (256, 256)), ("output_layer_rgb.0", (3, 256)),
    ],
    "tiny_nerf": [
        ("layers_pos.0", (256, 60)), ("layers_pos.1", (256, 256)), ("layers_pos.2", (256, 256)),
        ("layers_pos.3", (256, 256)),
        ("layers_dir.0", (128, 280)),
        ("output_layer_sigma", (1, 256)), ("output_layer_rgb", (3, 128)),
    ],
}

MACS = {k: sum(o * i for _, (o, i) in v) for k, v in SPECS.items()}
N_FILM_LAYERS = 9  # input + 7 hidden + rgb hidden (pi_GAN/modules.py:52-54: 8 + 1 mapping heads)


def param_shapes(kind: str) -> dict:
    out = {}
    for key, (o, i) in SPECS[kind]:
        out[key + ".weight"] = (o, i)
        out[key + ".bias"] = (o,)
    return out


def posenc(x: torch.Tensor, length: int) -> torch.Tensor:
    """nerf/nerf.py:44-49."""
    cols = []
    for i in range(length):
        s = 2.0 ** i
        cols += [torch.sin(s * x), torch.cos(s * x)]
    return torch.cat(cols, dim=-1)


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def nerf_forward(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """nerf/nerf.py:75-94."""
    pos, d = x[..., :3], x[..., 3:6]
    e_pos, e_dir = posenc(pos, 10), posenc(d, 4)
    h = e_pos
    for l in range(5):
        h = torch.relu(_lin(sd, f"layers_pos.{l}", h))
    h = torch.cat([e_pos, h], -1)
    for l in range(5, 8):
        h = torch.relu(_lin(sd, f"layers_pos.{l}", h))
    sigma = torch.relu(_lin(sd, "output_layer_sigma", h))
    h = _lin(sd, "layers_dir.0", h)
    h = torch.relu(_lin(sd, "layers_dir.1", torch.cat([h, e_dir], -1)))
    rgb = torch.sigmoid(_lin(sd, "output_layer_rgb", h))
    return torch.cat([rgb, sigma], -1)


def tiny_nerf_forward(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """Build-defined C1 network: PE -> 4 x Dense-ReLU(256) -> heads."""
    pos, d = x[..., :3], x[..., 3:6]
    e_pos, e_dir = posenc(pos, 10), posenc(d, 4)
    h = e_pos
    for l in range(4):
        h = torch.relu(_lin(sd, f"layers_pos.{l}", h))
    sigma = torch.relu(_lin(sd, "output_layer_sigma", h))
    h = torch.relu(_lin(sd, "layers_dir.0", torch.cat([h, e_dir], -1)))
    rgb = torch.sigmoid(_lin(sd, "output_layer_rgb", h))
    return torch.cat([rgb, sigma], -1)


def siren_nerf_forward(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """nerf/nerf.py:153-170; Siren layer nerf/nerf.py:111-112 = sin(30 * linear)."""
    pos, d = x[..., :3], x[..., 3:6]
    h = pos
    for l in range(5):
        h = torch.sin(W0 * _lin(sd, f"layers_pos.{l}", h))
    h = torch.cat([pos, h], -1)
    for l in range(5, 8):
        h = torch.sin(W0 * _lin(sd, f"layers_pos.{l}", h))
    sigma = torch.relu(_lin(sd, "output_layer_sigma", h))
    h = _lin(sd, "layers_dir.0", h)
    h = torch.sin(W0 * _lin(sd, "layers_dir.1", torch.cat([h, d], -1)))
    rgb = torch.sigmoid(_lin(sd, "output_layer_rgb", h))
    return torch.cat([rgb, sigma], -1)


def film_siren_nerf_forward(sd: dict, film: torch.Tensor, x: torch.Tensor, use_dir: bool = True) -> torch.Tensor:
    """pi_GAN/modules.py:101-118.  ``film`` is one image's mapping output [9, 512]:
    row l = (gamma[256] | beta[256]) (torch.chunk(.,2) at modules.py:96-99)."""
    pos, d = x[..., :3], x[..., 3:6]

    def film_layer(key, h, l):
        g, b = film[l, :256], film[l, 256:]
        return torch.sin(W0 * (g * _lin(sd, key, h) + b))

    h = film_layer("input_layer", pos, 0)
    for l in range(7):
        h = film_layer(f"hidden_layers.{l}", h, l + 1)
    sigma = torch.relu(_lin(sd, "output_layer_sigma.0", h))
    if use_dir:
        h = torch.cat([h, d], -1)
    h = film_layer("hidden_layer_rgb", h, 8)
    rgb = torch.sigmoid(_lin(sd, "output_layer_rgb.0", h))
    return torch.cat([rgb, sigma], -1)


def make_field(kind: str, sd: dict, film: torch.Tensor | None = None):
    """Return a callable ``f(x[M,6]) -> [M,4]`` closing over the weights."""
    if kind == "nerf":
        return lambda x: nerf_forward(sd, x)
    if kind == "tiny_nerf":
        return lambda x: tiny_nerf_forward(sd, x)
    if kind == "siren_nerf":
        return lambda x: siren_nerf_forward(sd, x)
    if kind == "film_siren_nerf":
        return lambda x: film_siren_nerf_forward(sd, film, x, True)
    if kind == "film_siren_nerf_nodir":
        return lambda x: film_siren_nerf_forward(sd, film, x, False)
    raise KeyError(kind)
