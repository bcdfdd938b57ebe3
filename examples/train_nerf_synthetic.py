#!/usr/bin/env python3
"""The loop of nerf/train_nerf.py (lines 78-98, 124-176) on the MI355X path, end to end and without a dataset.

No dataset ships with the reference (and none can be fetched), so the "photographs" are rendered here: a randomly
initialised teacher field is rendered from a ring of cameras with the product's own `render_image`, and a student pair
(coarse + fine) is fitted to those views with exactly the calls INTEGRATION.md describes - `RayBank` (the rays_rgba
table, built and shuffled on the device), `render_rays` (the drop-in, with autograd), `nerf_loss` (train_nerf.py:158-167
in one kernel) and `FusedAdam` (train_nerf.py:98 + the repack of the weight streams) - then the held-out view is
rendered and scored with the device-side `psnr` / `ssim` (nerf/test_nerf.py:102-105).  Only the product is imported:
nothing from oracle/.

    python examples/train_nerf_synthetic.py [--steps 800] [--size 32] [--views 8] [--siren]
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "msra-practice-project_amd"))
sys.path.insert(0, os.path.join(ROOT, "msra-practice-project_amd", "nerf"))       # provides `render`, like the scripts use it

import render  # noqa: E402  (the drop-in module: render_rays, render_image, ...)
from mirender import fields, metrics, train  # noqa: E402


def pose_on_ring(radius, theta_deg, phi_deg):
    """camera-to-world of a camera on a sphere looking at the origin (the convention of nerf/data_loader.py:39-51)."""
    t, p = math.radians(theta_deg), math.radians(phi_deg)
    trans = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], np.float32)
    rot_p = np.array([[1, 0, 0, 0], [0, math.cos(p), -math.sin(p), 0], [0, math.sin(p), math.cos(p), 0], [0, 0, 0, 1]], np.float32)
    rot_t = np.array([[math.cos(t), 0, -math.sin(t), 0], [0, 1, 0, 0], [math.sin(t), 0, math.cos(t), 0], [0, 0, 0, 1]], np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    return flip @ rot_t @ rot_p @ trans


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=800)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--siren", action="store_true", help="SirenNeRF students (use_siren of the reference's configs)")
    ap.add_argument("--use-alpha", action="store_true", help="add 0.1 x the opacity loss of train_nerf.py:160,165 (use_alpha)")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    assert torch.cuda.is_available(), "needs a ROCm device: there is no CPU path"
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    W = H = args.size
    focal, near, far, nc, nf = 1.3875 * W, 2.0, 6.0, 32, 32

    # the scene: a fixed random TinyNeRF with a denser sigma head (x8, +2: about a third of the frame is covered; a sparser
    # scene lets a ReLU NeRF at 5e-4 collapse to "empty space", as the reference's would), photographed from a ring of
    # cameras (+ one held-out view)
    teacher = fields.TinyNeRF().to(dev)
    with torch.no_grad():
        teacher.output_layer_sigma.weight.mul_(8.0)
        teacher.output_layer_sigma.bias.add_(2.0)
    poses = [pose_on_ring(4.0, 360.0 * k / args.views, -30.0) for k in range(args.views)] + [pose_on_ring(4.0, 23.0, -25.0)]
    shots = [render.render_image(W, H, focal, p, near, far, teacher, teacher, nc, 0, seed=100 + k) for k, p in enumerate(poses)]
    rgba = np.stack([np.concatenate([rgb, acc], -1) for rgb, _depth, acc in shots]).astype(np.float32)     # [N,H,W,4]

    # train_nerf.py:78-98
    # (the renderer composites on white itself, render.py:101, so the shots ARE the white-background pictures: white_bkgd=False
    #  keeps RayBank from compositing them a second time; alpha = the teacher's accumulated opacity)
    bank = train.RayBank(rgba[:-1], np.stack(poses[:-1]), focal, device=dev, white_bkgd=False)
    bank.shuffle()
    cls = fields.SirenNeRF if args.siren else fields.NeRF
    coarse, fine = cls().to(dev), cls().to(dev)
    lr0 = 1e-4 if args.siren else 5e-4
    opt = train.FusedAdam([coarse, fine], lr=lr0, betas=(0.9, 0.999))

    t0, first, last = time.time(), None, None
    for step in range(args.steps):                                          # train_nerf.py:124-176
        rays, rgb, alpha = bank.batch(args.batch)
        out = render.render_rays(rays, near, far, coarse, fine, nc, nf)
        loss, psnr = train.nerf_loss(out, rgb, alpha, use_alpha=args.use_alpha, use_fine_model=True)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g["lr"] = train.decayed_lr(lr0, 250, step + 1)
        if step == 0:
            first = float(loss)
        if not args.quiet and (step % 50 == 0 or step == args.steps - 1):
            print(f"step {step:4d}  loss {float(loss):.5f}  batch psnr {float(psnr):.2f} dB", flush=True)
        last = float(loss)
    torch.cuda.synchronize()
    dt = time.time() - t0

    # nerf/test_nerf.py:96-105: a training view rendered again (how well the pictures were fitted) and the held-out view (a
    # random field's view-dependent colour says little about a direction it was never seen from: informational)
    def score(k, seed):
        rgb, _depth, _acc = render.render_image(W, H, focal, poses[k], near, far, coarse, fine, nc, nf, seed=seed)
        got = torch.from_numpy(rgb).permute(2, 0, 1)[None].to(dev)
        want = torch.from_numpy(rgba[k][..., :3]).permute(2, 0, 1)[None].to(dev)
        return float(metrics.psnr(got, want)), float(metrics.ssim(got, want))
    fit_psnr, fit_ssim = score(0, 7)
    held_psnr, held_ssim = score(len(poses) - 1, 8)
    result = {"first_loss": first, "last_loss": last, "train_view_psnr_db": fit_psnr, "train_view_ssim": fit_ssim,
              "heldout_psnr_db": held_psnr, "heldout_ssim": held_ssim, "steps": args.steps,
              "rays_per_s": args.steps * args.batch / dt}
    if not args.quiet:
        print(f"{args.steps} steps of {args.batch} rays in {dt:.1f} s ({result['rays_per_s'] / 1e3:.0f} k rays/s incl. Python); "
              f"training view 0: PSNR {fit_psnr:.2f} dB, SSIM {fit_ssim:.3f}; held-out view: {held_psnr:.2f} dB, {held_ssim:.3f}")
    return result


if __name__ == "__main__":
    main()
