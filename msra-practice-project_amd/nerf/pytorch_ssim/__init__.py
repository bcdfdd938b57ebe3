"""Drop-in for the reference's nerf/pytorch_ssim package (`import pytorch_ssim` in nerf/test_nerf.py:10,
used at :104): same `ssim`, `SSIM`, `gaussian`, `create_window`, evaluated by one HIP kernel on the device."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirender.metrics import SSIM, create_window, gaussian, ssim  # noqa: E402,F401
