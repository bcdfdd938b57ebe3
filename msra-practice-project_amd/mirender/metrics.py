"""Frame metrics on the device: mse / psnr (nerf/test_nerf.py:102-103, train_nerf.py:160) and pytorch_ssim.ssim
(nerf/pytorch_ssim/__init__.py), one C-ABI call, no host round trip until the caller asks for `.item()`.

Same names and argument meaning as the reference's pytorch_ssim module: `ssim(img1, img2, window_size=11,
size_average=True)`, `SSIM(window_size, size_average)`; images are NCHW float tensors on a ROCm device.
"""
from __future__ import annotations

import ctypes
from math import exp

import torch

from . import _lib


def gaussian(window_size: int, sigma: float) -> torch.Tensor:
    """pytorch_ssim.gaussian (__init__.py:7-10).  Always a CPU tensor, whatever the ambient default tensor type
    (the scripts set torch.cuda.FloatTensor, nerf/test_nerf.py:13): mi_image_metrics takes the window as a HOST array."""
    gauss = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)],
                         dtype=torch.float32, device="cpu")
    return gauss / gauss.sum()


def create_window(window_size: int, channel: int) -> torch.Tensor:
    """pytorch_ssim.create_window (__init__.py:12-16): [channel,1,ws,ws] outer-product window (CPU)."""
    g = gaussian(window_size, 1.5).unsqueeze(1)
    return g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()


def image_metrics(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """[N,2] device tensor: per image (mse, mean ssim) of img1 vs img2, both [N,C,H,W] fp32 on the device."""
    lib = _lib.load()
    if img1.dim() != 4 or img1.shape != img2.shape:
        raise _lib.MiRenderError("image_metrics expects two [N,C,H,W] tensors of the same shape")
    if not img1.is_cuda or img2.device != img1.device:
        raise _lib.MiRenderError("image_metrics needs both images on the same ROCm device (there is no CPU path)")
    dev = img1.device
    a = img1.detach().to(torch.float32).contiguous()
    b = img2.detach().to(torch.float32).contiguous()
    n, c, h, w = a.shape
    win = gaussian(window_size, 1.5).contiguous()
    if win.device.type != "cpu" or win.dtype != torch.float32:
        raise _lib.MiRenderError("the SSIM window must be a CPU fp32 tensor (mi_image_metrics reads it on the host)")
    ws = torch.empty(lib.mi_image_metrics_workspace_floats(n, c, h, w), dtype=torch.float32, device=dev)
    out = torch.empty((n, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mi_image_metrics(_lib.ptr(a), _lib.ptr(b), n, c, h, w,
                                        ctypes.c_void_p(win.data_ptr()), int(window_size), _lib.ptr(ws), _lib.ptr(out),
                                        _lib.stream_ptr(dev)), "mi_image_metrics")
    return out


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    """pytorch_ssim.ssim (__init__.py:66-73): scalar tensor, or one value per image when size_average=False."""
    m = image_metrics(img1, img2, window_size)[:, 1]
    return m.mean() if size_average else m


def mse(img1, img2):
    """torch.mean((image - target)**2) over the whole batch (test_nerf.py:102)."""
    return image_metrics(img1, img2)[:, 0].mean()


def psnr(img1, img2):
    """-10 log10(mse) (test_nerf.py:103, train_nerf.py:160)."""
    return -10.0 * torch.log10(mse(img1, img2))


class SSIM(torch.nn.Module):
    """pytorch_ssim.SSIM (__init__.py:39-64)."""

    def __init__(self, window_size: int = 11, size_average: bool = True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.size_average)
