"""The data path of nerf/train_nerf.py's loop around render_rays, on the device:

  RayBank      the rays_rgba table and its batches         train_nerf.py:78-86, 143-147
  nerf_loss    loss + psnr of one batch, one kernel        train_nerf.py:158-167
  decayed_lr   the learning-rate schedule                  train_nerf.py:170-175

  FusedAdam    torch.optim.Adam's step + the repack of both   train_nerf.py:98, 168
               MFMA weight streams, one launch
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import _lib, fields


class RayBank:
    """rays_rgba [N*H*W, 10] = (rays_o, rays_d, r, g, b, a) for every pixel of every training image, built and
    shuffled on the device.  images [N,H,W,4] (RGBA in [0,1]), poses [N,>=3,4] camera-to-world, focal a float.

    By default the per-epoch reshuffle takes effect; the reference's does not (train_nerf.py:144 assigns the permuted
    table to a misspelt name, so every epoch replays the first shuffle): `reshuffle=False` reproduces that for
    trajectory comparisons.  `focal` may be a Python float or - as nerf/data_loader.py:151 returns it - an np.float64
    scalar, in which case NumPy >= 2 evaluates get_rays in fp64 (train_nerf.py:78) and the table follows suit."""

    def __init__(self, images, poses, focal: float, device="cuda", white_bkgd: bool = True, generator=None,
                 reshuffle: bool = True):
        lib = _lib.load()
        dev = torch.device(device)
        images = torch.as_tensor(np.asarray(images) if not isinstance(images, torch.Tensor) else images)
        images = images.to(device=dev, dtype=torch.float32).contiguous()
        n, h, w, c = images.shape
        if c != 4:
            raise _lib.MiRenderError("RayBank expects RGBA images [N,H,W,4]")
        poses = torch.as_tensor(np.asarray(poses) if not isinstance(poses, torch.Tensor) else poses)
        poses = poses.to(dtype=torch.float32)[:, :3, :4].contiguous().reshape(n, 12).to(dev)
        self.width, self.height, self.generator, self.reshuffle = w, h, generator, reshuffle
        self.table = torch.empty((n * h * w, 10), dtype=torch.float32, device=dev)
        f64 = isinstance(focal, np.floating) and np.dtype(type(focal)) == np.float64      # as ops.gen_rays
        with torch.cuda.device(dev):
            _lib.check(lib.mi_ray_bank(w, h, float(focal), _lib.ptr(poses), _lib.ptr(images), int(white_bkgd), n,
                                       _lib.ptr(self.table), int(f64), _lib.stream_ptr(dev)), "mi_ray_bank")
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
            if self.reshuffle:
                self.shuffle()
            self.batch_idx = 0
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
        g = torch._foreach_mul(list(ctx.saved_tensors), g_loss)          # one launch for the four seeds
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


class FusedAdam:
    """torch.optim.Adam(params, lr, betas) of nerf/train_nerf.py:98 for the parameters of one or two fused field
    modules, with the same interface the script uses - `zero_grad()`, `step()`, `param_groups[i]['lr'] = ...`
    (train_nerf.py:174-175), `state_dict()` / `load_state_dict()` in torch.optim.Adam's own format, so the
    'optimizer' entry of a checkpoint moves freely between the two (train_nerf.py:110,187).

    One C-ABI call per step (mi_adam_step): every tensor's Adam update AND the refresh of both packed MFMA weight
    streams of each model in a single launch; a torch optimiser step is followed by two pack launches per model at
    the next forward / backward.  Arithmetic follows torch's multi-tensor Adam op by op (parity 1e-7 over 10 steps:
    tests/test_gpu_trainloop.py)."""

    def __init__(self, models, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        models = [models] if isinstance(models, torch.nn.Module) else list(models)
        if not 1 <= len(models) <= 2:
            raise _lib.MiRenderError("FusedAdam takes one or two field modules (coarse, fine)")
        self.fields = []
        for m in models:
            pf = fields.as_packed_field(m)
            if pf is None:
                raise _lib.MiRenderError("FusedAdam needs modules with a known field layout (fields.detect_kind)")
            if pf not in self.fields:                      # coarse_model is fine_model (use_fine_model off): one field
                self.fields.append(pf)
        self.params = [p for pf in self.fields for p in pf.params]
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)]
        self.state = {}
        self._step = 0

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def _ensure_state(self):
        for p in self.params:
            if p not in self.state:
                self.state[p] = dict(step=torch.tensor(float(self._step), device="cpu"), exp_avg=torch.zeros_like(p),
                                     exp_avg_sq=torch.zeros_like(p))

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        g = self.param_groups[0]
        lr, (beta1, beta2), eps = float(g["lr"]), g["betas"], float(g["eps"])
        if g.get("weight_decay") or g.get("amsgrad") or g.get("maximize"):
            raise _lib.MiRenderError("FusedAdam implements plain Adam (train_nerf.py:98): weight_decay / amsgrad / maximize "
                                     "are not supported - use torch.optim.Adam for those")
        if any(p.grad is None for p in self.params):
            raise _lib.MiRenderError("FusedAdam.step: every parameter needs a gradient (run backward first)")
        self._ensure_state()
        self._step += 1
        t = self._step
        bc1 = 1 - beta1 ** t
        bc2_sqrt = math.sqrt(1 - beta2 ** t)
        n = len(self.params)
        for pf in self.fields:
            pf._follow_device()                    # `model.to(other_device)` after construction re-homes the streams
        dev = self.fields[0].device
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in self.params]
        for p, gr in zip(self.params, grads):
            if not p.is_contiguous() or gr.dtype != torch.float32 or p.device != dev or gr.device != dev:
                raise _lib.MiRenderError("FusedAdam: parameters and gradients must be contiguous fp32 on one device "
                                         f"({dev}); got a parameter on {p.device} with its gradient on {gr.device}")
            st = self.state[p]
            if st["exp_avg"].device != dev:        # the moments follow the parameters (raw pointers go to the kernel)
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].to(dev), st["exp_avg_sq"].to(dev)
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])  # noqa: E731
        kinds = (ctypes.c_int * len(self.fields))(*[pf.kind for pf in self.fields])
        numel = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        # the streams must hold the current parameters before their changed entries are scattered into them
        fwd = [pf.refresh() for pf in self.fields]
        bwd = [pf.refresh_bwd() if pf.packed_bwd is not None else None for pf in self.fields]
        pfwd = (ctypes.c_void_p * len(fwd))(*[x.data_ptr() for x in fwd])
        pbwd = (ctypes.c_void_p * len(bwd))(*[None if x is None else x.data_ptr() for x in bwd])
        with torch.cuda.device(dev):
            _lib.check(lib.mi_adam_step(len(self.fields), kinds, arr(self.params), arr(grads),
                                        arr([self.state[p]["exp_avg"] for p in self.params]),
                                        arr([self.state[p]["exp_avg_sq"] for p in self.params]), numel,
                                        -lr / bc1, 1 - beta1, beta2, 1 - beta2, eps, bc2_sqrt, pfwd, pbwd,
                                        _lib.stream_ptr(dev)), "mi_adam_step")
        for p in self.params:
            self.state[p]["step"] += 1
        # the kernel wrote parameters and streams together; torch's version counters did not move, so the lazily
        # refreshed streams stay valid (re-stamped), while the field's own epoch tells a forward that is still waiting
        # for its backward that the weights changed under it (autograd._RenderRaysFn.backward raises, as PyTorch does
        # for its own saved tensors).  A stream that did not exist yet is built from the parameters on first use.
        for pf in self.fields:
            pf.note_fused_update()

    # -- torch.optim.Adam's state-dict format ------------------------------------------------------------------
    def state_dict(self):
        self._ensure_state() if self._step else None
        state = {i: {k: v.clone() for k, v in self.state[p].items()} for i, p in enumerate(self.params) if p in self.state}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = dict(sd["param_groups"][0])
        if len(group["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameters")
        for k in ("lr", "betas", "eps"):
            self.param_groups[0][k] = tuple(group[k]) if k == "betas" else group[k]
        self.state = {}
        steps = set()
        for i, st in sd["state"].items():
            p = self.params[int(i)]
            self.state[p] = dict(step=torch.as_tensor(st["step"], dtype=torch.float32).clone().cpu(),
                                 exp_avg=st["exp_avg"].to(device=p.device, dtype=torch.float32).clone().contiguous(),
                                 exp_avg_sq=st["exp_avg_sq"].to(device=p.device, dtype=torch.float32).clone().contiguous())
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdam keeps one step count for all parameters")
        self._step = steps.pop() if steps else 0
