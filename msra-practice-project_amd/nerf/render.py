"""Drop-in for the reference's nerf/render.py: put this directory on sys.path ahead of the
reference's and `from render import *` (nerf/train_nerf.py:8, test_nerf.py:8, show_nerf.py:5,
demo_view.py:7, demo_param.py:7) picks up the MI355X path with the same names and signatures.

Also leaks `torch`, `np`, `tqdm` like the reference module does (scripts rely on the star import).
"""
import os
import sys

import numpy as np  # noqa: F401
import torch  # noqa: F401
from tqdm import tqdm  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirender.render_core import (  # noqa: E402,F401
    get_rays, raw_to_outputs, render_image, render_rays, render_video, run_network, sample_pdf, to8b)
