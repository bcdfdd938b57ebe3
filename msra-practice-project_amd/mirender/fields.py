"""Radiance-field models on the fused HIP MLP kernel.

Two jobs:

1. Recognise the reference's model objects (nerf/nerf.py NeRF / SirenNeRF, pi_GAN/modules.py
   FilmSirenNeRF) by their parameter layout, so `render_rays(rays, near, far, coarse_model,
   fine_model, ...)` keeps the reference call surface (nerf/render.py:106) while the
   `network(inputs)` call of run_network (render.py:72-74) runs in one fused kernel.
   Parameters are read AT CALL TIME (the optimiser mutates them in place) and repacked into
   the MFMA-ordered stream only when a parameter's version counter has changed.

2. Provide the same model families as nn.Modules with identical state-dict keys
   (checkpoints of the reference load with load_state_dict) whose forward([M,6]) -> [M,4]
   is the fused kernel.  There is no PyTorch/CPU forward here on purpose: without the HIP
   library these modules raise.
"""
from __future__ import annotations

import ctypes
import math
import warnings
import weakref

import numpy as np
import torch

from . import _lib

NERF, SIREN_NERF, FILM_SIREN_NERF, FILM_SIREN_NERF_NODIR, TINY_NERF = range(5)
KIND_NAMES = {NERF: "nerf", SIREN_NERF: "siren_nerf", FILM_SIREN_NERF: "film_siren_nerf",
              FILM_SIREN_NERF_NODIR: "film_siren_nerf_nodir", TINY_NERF: "tiny_nerf"}

# (key, (out, in)) per linear layer in the order the C ABI expects (mi_field_pack)
_HID = [(f"hidden_layers.{i}", (256, 256)) for i in range(7)]
SPECS = {
    NERF: [("layers_pos.0", (256, 60))] + [(f"layers_pos.{i}", (256, 256)) for i in (1, 2, 3, 4)]
          + [("layers_pos.5", (256, 316)), ("layers_pos.6", (256, 256)), ("layers_pos.7", (256, 256)),
             ("layers_dir.0", (256, 256)), ("layers_dir.1", (128, 280)),
             ("output_layer_sigma", (1, 256)), ("output_layer_rgb", (3, 128))],
    SIREN_NERF: [("layers_pos.0", (256, 3))] + [(f"layers_pos.{i}", (256, 256)) for i in (1, 2, 3, 4)]
                + [("layers_pos.5", (256, 259)), ("layers_pos.6", (256, 256)), ("layers_pos.7", (256, 256)),
                   ("layers_dir.0", (256, 256)), ("layers_dir.1", (128, 259)),
                   ("output_layer_sigma", (1, 256)), ("output_layer_rgb", (3, 128))],
    FILM_SIREN_NERF: [("input_layer", (256, 3))] + _HID
                     + [("output_layer_sigma.0", (1, 256)), ("hidden_layer_rgb", (256, 259)),
                        ("output_layer_rgb.0", (3, 256))],
    FILM_SIREN_NERF_NODIR: [("input_layer", (256, 3))] + _HID
                           + [("output_layer_sigma.0", (1, 256)), ("hidden_layer_rgb", (256, 256)),
                              ("output_layer_rgb.0", (3, 256))],
    TINY_NERF: [("layers_pos.0", (256, 60))] + [(f"layers_pos.{i}", (256, 256)) for i in (1, 2, 3)]
               + [("layers_dir.0", (128, 280)), ("output_layer_sigma", (1, 256)), ("output_layer_rgb", (3, 128))],
}
MACS = {k: sum(o * i for _, (o, i) in v) for k, v in SPECS.items()}
FLOPS_PER_POINT = {k: 2 * v for k, v in MACS.items()}   # 2 x MACs of the linear layers


def is_film(kind: int) -> bool:
    return kind in (FILM_SIREN_NERF, FILM_SIREN_NERF_NODIR)


def _shapes(kind):
    out = {}
    for key, (o, i) in SPECS[kind]:
        out[key + ".weight"] = (o, i)
        out[key + ".bias"] = (o,)
    return out


def detect_kind(named_params: dict) -> int | None:
    """Match a {name: tensor} mapping against the known layouts (exact key set and shapes)."""
    got = {k: tuple(v.shape) for k, v in named_params.items()}
    for kind in SPECS:
        if got == _shapes(kind):
            return kind
    return None


W_0 = 30.0     # Siren's hard-coded frequency (nerf/nerf.py:112) and FilmSiren's default (pi_GAN/modules.py:11).  The FiLM
#                kernels take the module's own w_0 at run time (one value per network, as FilmSirenNeRF passes one to every
#                layer, modules.py:73-94): it travels in the packed stream (mi_field_pack).


def expected_activation(kind: int, key: str) -> str:
    """What the fused kernel of `kind` applies after layer `key`: "relu" | "linear" | "sigmoid" (Dense's names,
    nerf/nerf.py:15-16), "sin" (Siren, nerf/nerf.py:111-112) or "film" (FilmSiren, pi_GAN/modules.py:22-25)."""
    if key.startswith("output_layer_sigma"):
        return "relu"
    if key.startswith("output_layer_rgb"):
        return "sigmoid"
    if is_film(kind):
        return "film"
    if key == "layers_dir.0" and kind != TINY_NERF:
        return "linear"
    return "sin" if kind == SIREN_NERF else "relu"


def _layer_module(model, key):
    mod = model
    for part in key.split("."):
        mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
    return mod


def film_w0(model, kind: int) -> float:
    """The w_0 the fused kernels run a recognised module with: its first FiLM layer's `w_0` attribute (the reference's
    FilmSirenNeRF gives every layer the same one, pi_GAN/modules.py:73-94; hyper_mismatch refuses a module whose layers
    disagree), our own modules' `w_0`, 30 for everything else."""
    if not is_film(kind):
        return W_0
    if hasattr(model, "mi_w_0"):
        return float(model.mi_w_0)
    try:
        return float(getattr(_layer_module(model, SPECS[kind][0][0]), "w_0", W_0))
    except (AttributeError, IndexError, KeyError, TypeError):
        return W_0


def hyper_mismatch(model, kind: int) -> str | None:
    """A layout match is not enough: the kernels hard-code each layer's activation and w_0 = 30.  Walk the layer
    objects of a recognised module and compare what they say about themselves with what the kernel of `kind`
    computes; returns a description of the first mismatch (the caller then takes the generic path, which calls the
    module's own forward), or None.

      Dense       (nerf/nerf.py:5-28)        carries `activation_name`
      FilmSiren   (pi_GAN/modules.py:8-31)   carries `w_0` (and is only right for the "film" slots)
      Siren       (nerf/nerf.py:97-117)      carries neither: recognised by its class name, sin(30 .) hard-coded there
      nn.Linear inside nn.Sequential(Linear, ReLU|Sigmoid)  (modules.py:81-84,89-92): the sibling names the activation
      our own leaves (`_Leaf`)               carry `mi_activation`
    Anything else (a user's nn.Linear stack with its own forward, a subclass with another nonlinearity) says nothing
    about its activation, so it is not claimed."""
    for key, _ in SPECS[kind]:
        want = expected_activation(kind, key)
        try:
            mod = _layer_module(model, key)
        except (AttributeError, IndexError, KeyError, TypeError):
            return f"{key}: no such layer object"
        if hasattr(mod, "mi_activation"):
            have = mod.mi_activation
        elif hasattr(mod, "activation_name"):
            have = mod.activation_name
        elif hasattr(mod, "w_0"):
            w = float(mod.w_0)
            if not (w > 0.0 and math.isfinite(w)):
                return f"{key}: w_0 = {mod.w_0} (the fused kernels need a finite w_0 > 0)"
            if w != film_w0(model, kind):
                return f"{key}: w_0 = {mod.w_0} differs from the first FiLM layer's (the fused kernels take one w_0 per network)"
            have = "film"
        elif type(mod).__name__ == "Siren":
            have = "sin"
        elif type(mod) is torch.nn.Linear and key.endswith(".0"):
            try:
                sib = _layer_module(model, key[:-2])[1]
            except (AttributeError, IndexError, KeyError, TypeError):
                return f"{key}: a bare Linear with no activation module after it"
            have = {"ReLU": "relu", "Sigmoid": "sigmoid"}.get(type(sib).__name__, type(sib).__name__)
        else:
            return f"{key}: {type(mod).__name__} does not name its activation"
        if have != want:
            return f"{key}: activation {have!r}, the fused {KIND_NAMES[kind]} kernel applies {want!r}"
    return None


class PackedField:
    """Packed MFMA-ordered weights of one model, refreshed lazily from its live parameters."""

    def __init__(self, kind: int, params: list, w_0: float = W_0):
        self.kind = kind
        self.params = params                      # live tensors, state-dict order (w0,b0,w1,b1,...)
        self.w_0 = float(w_0)                     # FiLM kinds: the module's w_0, packed into the streams' trailer
        self.device = None
        self._epoch = 0                           # bumped by writers that bypass the version counters (FusedAdam)
        self._follow_device()

    def _follow_device(self):
        """(Re)allocate the packed streams on the parameters' device: `model.to(other_device)` replaces the
        parameters' storage in place, so the same PackedField must move with them."""
        dev = self.params[0].device
        if dev == self.device:
            return
        if dev.type != "cuda":
            raise _lib.MiRenderError("the fused renderer needs the model on a ROCm device (model.cuda())")
        self.device = dev
        self.packed = torch.empty(_lib.load().mi_field_packed_floats(self.kind), dtype=torch.float32, device=dev)
        self.packed_bwd = None
        self._versions = self._versions_bwd = None

    def versions(self):
        """(storage pointer, version counter) per parameter: what decides whether the packed streams are stale.
        In-place updates through torch ops (optimiser steps, load_state_dict, `p.mul_()` under no_grad) bump the
        counter; writes through `.data` (the reference's `bias.data[:n] = 1`, pi_GAN/modules.py:57-58) do NOT -
        call invalidate() after such a write."""
        return (self._epoch, self.w_0) + tuple((p.data_ptr(), p._version) for p in self.params)

    def note_fused_update(self):
        """A kernel rewrote the parameters AND every existing packed stream consistently (mirender.train.FusedAdam):
        the streams stay valid - they are re-stamped - but anything that remembered `versions()` from before (a forward
        whose backward has not run yet: autograd._RenderRaysFn) must see that the weights moved."""
        self._epoch += 1
        vers = self.versions()
        if self._versions is not None:
            self._versions = vers
        if self._versions_bwd is not None:
            self._versions_bwd = vers

    def invalidate(self):
        """Force a repack on the next use (after writes the version counters cannot see, e.g. through `.data`)."""
        self._versions = self._versions_bwd = None

    def _sources(self):
        srcs = []
        for p in self.params:
            if p.dtype != torch.float32 or p.device != self.device:
                raise _lib.MiRenderError("field parameters must be fp32 on one device")
            srcs.append(p.detach() if p.is_contiguous() else p.detach().contiguous())
        return srcs

    def refresh_bwd(self):
        """Transposed weight stream for the backward chain (dX = W^T dA), same lazy refresh."""
        self._follow_device()
        vers = self.versions()
        if vers == self._versions_bwd:
            return self.packed_bwd
        lib = _lib.load()
        if self.packed_bwd is None:
            self.packed_bwd = torch.empty(lib.mi_field_packed_bwd_floats(self.kind), dtype=torch.float32,
                                          device=self.device)
        srcs = self._sources()
        arr = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        with torch.cuda.device(self.device):
            _lib.check(lib.mi_field_pack_bwd(self.kind, arr, len(srcs), self.w_0, _lib.ptr(self.packed_bwd),
                                             _lib.stream_ptr(self.device)), "mi_field_pack_bwd")
        self._keep_bwd = srcs
        self._versions_bwd = vers
        return self.packed_bwd

    def refresh(self):
        self._follow_device()
        vers = self.versions()
        if vers == self._versions:
            return self.packed
        lib = _lib.load()
        srcs = self._sources()
        arr = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        with torch.cuda.device(self.device):
            _lib.check(lib.mi_field_pack(self.kind, arr, len(srcs), self.w_0, _lib.ptr(self.packed),
                                         _lib.stream_ptr(self.device)), "mi_field_pack")
        self._keep = srcs
        self._versions = vers
        return self.packed


_field_cache: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _replica_named(model) -> dict:
    """torch.nn.DataParallel (pi_GAN/train.py:50) runs forward on per-device REPLICAS whose parameters are plain
    tensor attributes (broadcast copies that still carry autograd history back to the originals), so
    named_parameters() is empty there.  Walk the known layouts' attribute paths instead."""
    for kind, spec in SPECS.items():
        named = {}
        try:
            for key, _ in spec:
                mod = model
                for part in key.split("."):
                    mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
                named[key + ".weight"], named[key + ".bias"] = mod.weight, mod.bias
        except (AttributeError, IndexError, KeyError, TypeError):
            continue
        if all(isinstance(t, torch.Tensor) for t in named.values()) and detect_kind(named) == kind:
            return named
    return {}


def as_packed_field(model) -> PackedField | None:
    """PackedField for a recognised nn.Module (reference class or ours), else None."""
    if isinstance(model, PackedField):
        return model
    if not isinstance(model, torch.nn.Module):
        return None
    pf = _field_cache.get(model)
    if pf is None:
        named = dict(model.named_parameters())
        if not named and getattr(model, "_is_replica", False):
            named = _replica_named(model)
        kind = detect_kind(named)
        why = None if kind is None else hyper_mismatch(model, kind)
        if why is not None:
            # same layout, other arithmetic (w_0 != 30, another activation, layers that do not say): generic path - the
            # module's own forward between the sampling / compositing kernels, correct but orders of magnitude slower
            # than the fused kernels.  Said once per model (the verdict is cached and NOT re-checked if a layer's
            # attributes change later: INTEGRATION.md)
            warnings.warn(f"{type(model).__name__} has the parameter layout of {KIND_NAMES[kind]} but is not claimed by the "
                          f"fused kernels ({why}); it is rendered through the generic path (its own forward)",
                          RuntimeWarning, stacklevel=3)
            kind = None
        if kind is None:
            _field_cache[model] = False
            return None
        if not next(iter(named.values())).is_cuda:
            raise _lib.MiRenderError("the fused renderer needs the model on a ROCm device (model.cuda())")
        params = []
        for key, _ in SPECS[kind]:
            params += [named[key + ".weight"], named[key + ".bias"]]
        pf = PackedField(kind, params, film_w0(model, kind))
        _field_cache[model] = pf
    return pf or None


def film_table(model) -> torch.Tensor:
    """[1,9,512] FiLM table from FilmSirenNeRF.film_params (list of 9 (gamma, beta) pairs set by
    set_film_params, pi_GAN/modules.py:96-99).  Raises ValueError like the reference
    (modules.py:106-107) when unset."""
    fp = getattr(model, "film_params", None)
    if fp is None:
        raise ValueError
    if isinstance(fp, torch.Tensor):
        return fp.reshape(-1, 9, 512)
    return torch.cat([torch.cat([g.reshape(-1), b.reshape(-1)]) for g, b in fp]).reshape(1, 9, 512)


def eval_points(pf: PackedField, x: torch.Tensor, film: torch.Tensor | None = None) -> torch.Tensor:
    """network(x[M,6]) -> [M,4] on the fused kernel (replaces the model call at render.py:73)."""
    lib = _lib.load()
    if x.dim() != 2 or x.shape[1] != 6:
        raise _lib.MiRenderError(f"expected inputs [M,6], got {tuple(x.shape)}")
    x = x.detach().to(device=pf.device, dtype=torch.float32).contiguous()
    out = torch.empty((x.shape[0], 4), dtype=torch.float32, device=pf.device)
    groups, ppg = 1, x.shape[0]
    if is_film(pf.kind):
        if film is None:
            raise ValueError
        film = film.detach().to(device=pf.device, dtype=torch.float32).contiguous().reshape(-1, 9, 512)
        groups = film.shape[0]
        if x.shape[0] % groups:
            raise _lib.MiRenderError("points must split evenly over the FiLM groups")
        ppg = x.shape[0] // groups
    with torch.cuda.device(pf.device):
        _lib.check(lib.mi_field_eval_points(pf.kind, _lib.ptr(pf.refresh()), _lib.ptr(film), _lib.ptr(x), groups,
                                            ppg, _lib.ptr(out), _lib.stream_ptr(pf.device)), "mi_field_eval_points")
    return out


# ------------------------------------------------------------------------------------------
# nn.Module families with the reference's state-dict layout and initialisers
# ------------------------------------------------------------------------------------------
class _Leaf(torch.nn.Module):
    """One linear layer's parameters; `mi_activation` names what the fused kernel applies after it."""

    def __init__(self, activation: str):
        super().__init__()
        self.mi_activation = activation


class _FusedField(torch.nn.Module):
    KIND = None

    def __init__(self):
        super().__init__()
        for key, (o, i) in SPECS[self.KIND]:
            self._register(key, torch.nn.Parameter(torch.empty(o, i)), torch.nn.Parameter(torch.zeros(o)))
        self.reset_parameters()

    def _register(self, key, w, b):
        # "layers_pos.3" -> self.layers_pos (ModuleList, indexable like the reference's) [3]
        leaf = _Leaf(expected_activation(self.KIND, key))
        leaf.weight, leaf.bias = w, b
        parts = key.split(".")
        if len(parts) == 1:
            self.add_module(parts[0], leaf)
            return
        if not hasattr(self, parts[0]):
            self.add_module(parts[0], torch.nn.ModuleList())
        lst = getattr(self, parts[0])
        assert int(parts[1]) == len(lst)
        lst.append(leaf)

    def _layer(self, key):
        parts = key.split(".")
        mod = getattr(self, parts[0])
        return mod[int(parts[1])] if len(parts) > 1 else mod

    @property
    def flops_per_point(self):
        return FLOPS_PER_POINT[self.KIND]

    def forward(self, input_tensor):
        from . import autograd          # gradients reach the parameters like the reference module's own forward
        return autograd.field_eval_points(as_packed_field(self), input_tensor)


def _xavier(w, gain):
    torch.nn.init.xavier_uniform_(w, gain=gain)


class NeRF(_FusedField):
    """nerf/nerf.py:52-94 (PE(10)/PE(4), 8x256 ReLU, skip at layer 5)."""
    KIND = NERF

    def reset_parameters(self):
        # Dense.reset_parameters nerf/nerf.py:25-28: xavier_uniform with the activation's gain, zero bias
        for key, _ in SPECS[self.KIND]:
            act = "relu"
            if key == "output_layer_rgb":
                act = "sigmoid"
            if self.KIND == NERF and key == "layers_dir.0":
                act = "linear"
            _xavier(self._layer(key).weight, torch.nn.init.calculate_gain(act))
            torch.nn.init.zeros_(self._layer(key).bias)


class TinyNeRF(NeRF):
    """Build-defined 4-layer variant for BASELINE config C1."""
    KIND = TINY_NERF


class SirenNeRF(_FusedField):
    """nerf/nerf.py:120-170 (sin(30 .) layers on raw xyz / dir)."""
    KIND = SIREN_NERF

    def reset_parameters(self):
        for key, (o, i) in SPECS[self.KIND]:
            lay = self._layer(key)
            if key in ("layers_dir.0", "output_layer_rgb"):
                _xavier(lay.weight, 1.0)
            elif key == "output_layer_sigma":
                _xavier(lay.weight, math.sqrt(2.0))
            else:  # Siren.reset_parameters nerf/nerf.py:114-117, first layer nerf/nerf.py:134
                bound = 1 / 30 if key == "layers_pos.0" else math.sqrt(6 / i) / 30
                torch.nn.init.uniform_(lay.weight, -bound, bound)
            torch.nn.init.zeros_(lay.bias)


class FilmSirenNeRF(_FusedField):
    """pi_GAN/modules.py:70-118, same constructor arguments (:73).  FiLM parameters are per-image state set by
    set_film_params (modules.py:96-99) or passed to forward, as in the reference.  `c` and `w_0` shape the initialisation
    (modules.py:27-31) and w_0 is the frequency the fused kernels run with; the fused kernels exist for the reference's own
    topology (hidden_dim 256, 8 layers) - other sizes have no kernel here and raise."""
    KIND = FILM_SIREN_NERF

    def __new__(cls, hidden_dim=256, hidden_layers=8, c=6, w_0=30, use_dir=True):
        if cls is FilmSirenNeRF and not use_dir:
            return super().__new__(FilmSirenNeRFNoDir)
        return super().__new__(cls)

    def __init__(self, hidden_dim=256, hidden_layers=8, c=6, w_0=30, use_dir=True):
        if (hidden_dim, hidden_layers) != (256, 8):
            raise _lib.MiRenderError(f"FilmSirenNeRF(hidden_dim={hidden_dim}, hidden_layers={hidden_layers}): the fused kernels "
                                     "implement the reference's default topology (256, 8) only; build the reference's own "
                                     "module for other sizes - render_rays drives it through the generic path")
        if not (float(w_0) > 0.0 and math.isfinite(float(w_0))):
            raise _lib.MiRenderError(f"FilmSirenNeRF(w_0={w_0}): need a finite w_0 > 0")
        self.c, self.mi_w_0 = c, float(w_0)       # before super().__init__: reset_parameters reads them
        super().__init__()
        self.w_0 = w_0
        self.use_dir = use_dir
        self.film_params = None

    def reset_parameters(self):
        for key, (o, i) in SPECS[self.KIND]:
            lay = self._layer(key)
            if key in ("output_layer_sigma.0", "output_layer_rgb.0"):   # torch.nn.Linear default
                b = 1 / math.sqrt(i)
                torch.nn.init.uniform_(lay.weight, -b, b)
                torch.nn.init.uniform_(lay.bias, -b, b)
            else:  # FilmSiren.reset_parameters pi_GAN/modules.py:27-31
                wb = 1 / i if key == "input_layer" else math.sqrt(self.c / i) / self.mi_w_0
                torch.nn.init.uniform_(lay.weight, -wb, wb)
                torch.nn.init.uniform_(lay.bias, -math.sqrt(1 / i), math.sqrt(1 / i))

    def set_film_params(self, mapping_tensor):
        self.film_params = [torch.chunk(mapping_tensor[i], 2) for i in range(mapping_tensor.shape[0])]

    def forward(self, input_tensor, film_params=None):
        if film_params is not None:
            self.film_params = film_params
        elif self.film_params is None:
            raise ValueError
        from . import autograd          # ... and the FiLM parameters (pi_GAN/synthesis.py:83-107 optimises them directly)
        return autograd.field_eval_points(as_packed_field(self), input_tensor, film_table(self))


class FilmSirenNeRFNoDir(FilmSirenNeRF):
    KIND = FILM_SIREN_NERF_NODIR

    def __init__(self, hidden_dim=256, hidden_layers=8, c=6, w_0=30, use_dir=False):
        super().__init__(hidden_dim, hidden_layers, c, w_0, use_dir=False)


def field_from_state_dict(sd: dict, device="cuda", w_0: float = W_0) -> torch.nn.Module:
    """Build the matching fused module for a reference state dict (checkpoint interop).  `w_0`: FiLM kinds only - a state
    dict does not carry FilmSiren's constructor argument (pi_GAN/modules.py:11,73), so a checkpoint of a w_0 != 30 network
    needs it said."""
    kind = detect_kind(sd)
    if kind is None:
        raise _lib.MiRenderError("state dict does not match a known field layout")
    cls = {NERF: NeRF, SIREN_NERF: SirenNeRF, FILM_SIREN_NERF: FilmSirenNeRF,
           FILM_SIREN_NERF_NODIR: FilmSirenNeRFNoDir, TINY_NERF: TinyNeRF}[kind]
    m = cls(w_0=w_0) if is_film(kind) else cls()
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
                       for k, v in sd.items()})
    return m.to(device)
