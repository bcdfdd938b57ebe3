"""Drop-in for the reference's pi_GAN/render.py (star-imported by pi_GAN/modules.py:3 and train.py:8).

Same names as the reference module: camera helpers (angles in RADIANS here, pi_GAN/render.py:37-49;
the nerf data loader's variant takes degrees), get_rays, render_rays, render_image (returns a device
tensor [H,W,3] with the autograd graph, pi_GAN/render.py:195-206), render_image_np, render_video_np.
"""
import os
import sys

import numpy as np
import torch  # noqa: F401
from tqdm import tqdm  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirender.render_core import (  # noqa: E402,F401
    get_rays, raw_to_outputs, render_image_np, render_rays, render_video_np, run_network, sample_pdf, to8b)
from mirender.render_core import render_image_tensor as render_image  # noqa: E402,F401


def trans_t(t):
    """Translation along camera z (pi_GAN/render.py:6-11)."""
    m = np.eye(4, dtype=np.float32)
    m[2, 3] = t
    return m


def rot_phi(phi):
    """Pitch about x (pi_GAN/render.py:14-19)."""
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def rot_theta(th):
    """Yaw about y (pi_GAN/render.py:22-27)."""
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float32)


blender_coord = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)


def camera_pos_to_transform_matrix(radius, theta, phi):
    """Camera-to-world from spherical position, radians (pi_GAN/render.py:37-49)."""
    return rot_theta(theta) @ (rot_phi(phi) @ trans_t(radius))
