"""mirender - MI355X-native volumetric renderer behind the reference's `render` call surface.

    from mirender import ops, fields, render_core

Drop-in modules named `render` live in ../nerf/render.py and ../pi_GAN/render.py.
"""
from . import _lib, fields, ops, render_core  # noqa: F401

__all__ = ["_lib", "fields", "ops", "render_core"]
