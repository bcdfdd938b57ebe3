#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (build container only).

Run once, here:   PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py
Outputs tests/golden/*.npz (committed).  /root/reference does not exist on the GPU
box; nothing but this script ever reads it.  Fixtures are data only: inputs,
injected jitter, every intermediate and the outputs of the reference functions
(nerf/render.py, nerf/nerf.py, pi_GAN/render.py, pi_GAN/modules.py) on CPU fp32.

Fixture families:
  F1 rays_*        get_rays, both camera helpers
  F2 composite_*   raw_to_outputs on hand-made edge cases
  F3 pdf_*         sample_pdf edge cases, Nf in {0,1,128}
  F4 field_*       every field MLP on 257 points (+ sharp variants)
  F5 render_*      render_rays with all intermediates
  F6 pigan_grad    pi_GAN image + grads wrt FiLM table and field weights
  F7 nerf_grad     nerf training loss grads for one ray batch
  F8 metrics       pytorch_ssim.ssim / mse / psnr of synthetic frame pairs (nerf/test_nerf.py:102-104)
  F9 video         render_video / render_image / render_image_np over two poses; ref_layouts.json (make_r02)
  ref_layer_attrs.json   each layer object's class / activation_name / w_0 in the reference's field classes (make_r03)
  F10 fit_r03_*    training trajectories of the reference's own code on the teacher scene (make_r03_fit)
      fit_r04_*    the same for the headline NeRF class (nerf/nerf.py:52-94), Adam 5e-4 and plain SGD (--only-r04-fit)
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import synth  # noqa: E402
from oracle import render_ref as oref  # noqa: E402
from oracle.fields import SPECS  # noqa: E402


def load_reference():
    sys.path.insert(0, os.path.join(REF, "nerf"))
    import render as nerf_render  # noqa
    import nerf as nerf_models  # noqa
    import data_loader as nerf_data  # noqa
    torch.autograd.set_detect_anomaly(False)  # nerf/nerf.py:2 turns it on process-wide
    sys.path.pop(0)
    for m in ("render",):
        del sys.modules[m]
    sys.path.insert(0, os.path.join(REF, "pi_GAN"))
    import render as pigan_render  # noqa
    import modules as pigan_modules  # noqa
    sys.path.pop(0)
    return nerf_render, nerf_models, nerf_data, pigan_render, pigan_modules


@contextlib.contextmanager
def injected_rand(queue):
    """Replace torch.rand (render.py:131) by tensors popped from ``queue``."""
    orig = torch.rand

    def fake(shape, *a, **k):
        t = queue.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.clone()
    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = orig


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  {len(out)} arrays")


def ref_model(kind, nerf_models, pigan_modules, sd, film=None):
    if kind == "nerf":
        m = nerf_models.NeRF()
    elif kind == "siren_nerf":
        m = nerf_models.SirenNeRF()
    elif kind == "film_siren_nerf":
        m = pigan_modules.FilmSirenNeRF(use_dir=True)
    elif kind == "film_siren_nerf_nodir":
        m = pigan_modules.FilmSirenNeRF(use_dir=False)
    else:
        raise KeyError(kind)
    m.load_state_dict(sd, strict=True)
    if film is not None:
        m.set_film_params(film)
    return m


def sample_points(n, seed, scale=1.5):
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = rng.uniform(-scale, scale, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return torch.from_numpy(np.concatenate([pos, d.astype(np.float32)], -1))


def subsample_idx(n, k=512):
    return np.unique(np.linspace(0, n - 1, min(n, k)).astype(np.int64))


def grad_summary(named_grads):
    out = {}
    for name, g in named_grads:
        g = g.detach().cpu().numpy().reshape(-1)
        idx = subsample_idx(g.size)
        out[f"g.{name}.idx"] = idx
        out[f"g.{name}.val"] = g[idx]
        out[f"g.{name}.sum"] = np.float64(g.astype(np.float64).sum())
        out[f"g.{name}.l2"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
    return out


def synth_frames(n, c, h, w, seed, noise):
    """Smooth 'rendered' frames in [0,1] and a noisy copy (a stand-in for a target photograph)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    a = np.empty((n, c, h, w), np.float32)
    for i in range(n):
        for k in range(c):
            fx, fy, ph = rng.uniform(1, 9), rng.uniform(1, 9), rng.uniform(0, 6.28)
            a[i, k] = 0.5 + 0.35 * np.sin(fx * xx * 6.28 + ph) * np.cos(fy * yy * 6.28) + 0.1 * (xx > 0.5)
    b = np.clip(a + noise * rng.standard_normal(a.shape).astype(np.float32), 0, 1).astype(np.float32)
    return torch.from_numpy(np.clip(a, 0, 1)), torch.from_numpy(b)


def make_metrics():
    """F8: the reference's own pytorch_ssim package on CPU (nerf/pytorch_ssim/__init__.py), plus the mse / psnr
    expressions of nerf/test_nerf.py:102-103."""
    sys.path.insert(0, os.path.join(REF, "nerf"))
    import pytorch_ssim  # noqa
    sys.path.pop(0)
    out = {}
    cases = [("ragged", 2, 3, 37, 53, 0.05), ("frame", 1, 3, 100, 100, 0.02), ("tiny", 1, 1, 5, 7, 0.1),
             ("wide", 3, 3, 16, 130, 0.2)]
    for name, n, c, h, w, noise in cases:
        a, b = synth_frames(n, c, h, w, seed=len(name) * 17 + h, noise=noise)
        out[f"{name}.img1"], out[f"{name}.img2"] = a.numpy(), b.numpy()
        out[f"{name}.ssim"] = pytorch_ssim.ssim(a, b).numpy()
        out[f"{name}.ssim_per_image"] = pytorch_ssim.ssim(a, b, size_average=False).numpy()
        out[f"{name}.ssim_w7"] = pytorch_ssim.ssim(a, b, window_size=7).numpy()
        out[f"{name}.ssim_module"] = pytorch_ssim.SSIM()(a, b).numpy()
        mse = torch.mean((a - b) ** 2)
        out[f"{name}.mse"] = mse.numpy()
        out[f"{name}.psnr"] = np.float64(-10 * torch.log10(mse).item())
    a, _ = synth_frames(1, 3, 40, 40, seed=5, noise=0.0)
    out["same.img1"] = a.numpy()
    out["same.ssim"] = pytorch_ssim.ssim(a, a.clone()).numpy()
    flat = torch.full((1, 3, 33, 33), 0.25)
    out["flat.ssim_vs_half"] = pytorch_ssim.ssim(flat, torch.full((1, 3, 33, 33), 0.5)).numpy()
    save("metrics_f8", **out)


def trace_render(render_mod, rays, near, far, cm, fm, nc, nf, tr):
    """Re-run the reference's stages one by one to capture intermediates, then
    check they reproduce render_rays' own outputs exactly."""
    with injected_rand([tr]):
        outs = render_mod.render_rays(rays, near, far, cm, fm, nc, nf)
    # glue between the reference's stage functions comes from the oracle; the assert
    # below proves the staged run reproduces render_rays bit for bit.
    ro, rdd = rays[:, 0], rays[:, 1]
    vd = rdd / torch.norm(rdd, dim=-1, keepdim=True)
    zc, mids = oref.stratified_z(rays.shape[0], near, far, nc, tr)
    raw_c = render_mod.run_network(oref.points_on_rays(ro, rdd, zc), vd, cm)
    rc = render_mod.raw_to_outputs(raw_c, zc, rdd)
    zs = render_mod.sample_pdf(mids, rc[3][..., 1:-1], nf)
    zf = torch.sort(torch.cat([zc, zs], -1), -1).values
    raw_f = render_mod.run_network(oref.points_on_rays(ro, rdd, zf), vd, fm)
    rf = render_mod.raw_to_outputs(raw_f, zf, rdd)
    for a, b in zip(outs, (rc[0], rc[1], rc[2], rf[0], rf[1], rf[2])):
        assert torch.equal(a, b)
    return dict(rgb_c=outs[0], depth_c=outs[1], acc_c=outs[2], rgb_f=outs[3], depth_f=outs[4], acc_f=outs[5],
                z_coarse=zc, raw_c=raw_c, weights_c=rc[3], z_samples=zs, z_fine=zf, raw_f=raw_f,
                weights_f=rf[3])


def pick_rays_from(render_mod, W_, H_, focal_, pose, n_rays, seed):
    o_, d_ = render_mod.get_rays(W_, H_, focal_, pose)
    rays = np.stack([o_, d_], 2).reshape(-1, 2, 3).astype(np.float32)
    idx = np.random.Generator(np.random.PCG64(seed)).choice(rays.shape[0], n_rays, replace=False)
    return torch.from_numpy(rays[np.sort(idx)])


def sharp_tag(sharp, plain=""):
    return {False: plain, True: "_sharp", "medium": "_medium"}[sharp]


def make_r02(nr, nm, nd, pr, pm):
    """Round-2 additions (the round-1 fixtures above are untouched):
      F5 twins  the sharp render_rays cases again with the reference's plain initialisation ("") and with the
                "medium" density head (oracle/synth.py), where the hard 1e-4 gate is the one in force; FiLM fields at
                12+24 and at the 24+48 samples of BASELINE config C5
      F9 video  nerf/render.py:170-182 render_video over two poses (and so render_image / pi_GAN's render_image_np)
      layouts   {class: {parameter name: shape}} of the reference's model classes (ref_layouts.json)"""
    import json
    with torch.no_grad():
        pose = nd.camera_pos_to_transform_matrix(4.0, 63.0, -30.0)
        for kind, nc, nf, nrays, sharp in (
            ("nerf", 32, 0, 96, False), ("nerf", 64, 0, 96, False),
            ("nerf", 32, 0, 96, "medium"), ("nerf", 64, 0, 96, "medium"), ("nerf", 64, 128, 96, "medium"),
            ("siren_nerf", 64, 128, 64, "medium"),
        ):
            rays = pick_rays_from(nr, 100, 100, 1.3875 * 100, pose, nrays, seed=nc + nf)
            sd_c = synth.state_dict(kind, seed=20, sharp=sharp, bias_jitter=0.05)
            sd_f = synth.state_dict(kind, seed=21, sharp=sharp, bias_jitter=0.05)
            cm, fm = ref_model(kind, nm, pm, sd_c), ref_model(kind, nm, pm, sd_f)
            tr = synth.t_rand(nrays, nc, seed=123)
            t = trace_render(nr, rays, 2.0, 6.0, cm, fm, nc, nf, tr)
            save(f"render_f5_{kind}_{nc}_{nf}{sharp_tag(sharp)}", rays=rays, t_rand=tr, near=2.0, far=6.0,
                 digest_c=np.array(synth.digest(sd_c)), digest_f=np.array(synth.digest(sd_f)), **t)
        pose_g = pr.camera_pos_to_transform_matrix(1.0, 0.2, -0.1)
        focal_g = float(32 / 2 / np.tan(12 / 2 * np.pi / 180))
        o_, d_ = pr.get_rays(32, 32, focal_g, pose_g)
        rays_g = torch.from_numpy(np.stack([o_, d_], 2).reshape(-1, 2, 3).astype(np.float32)[::8].copy())
        for kind, nc, nf, sharp in (("film_siren_nerf", 12, 24, False), ("film_siren_nerf", 12, 24, "medium"),
                                    ("film_siren_nerf_nodir", 12, 24, "medium"),
                                    ("film_siren_nerf", 24, 48, "medium"), ("film_siren_nerf", 24, 48, True)):
            sd = synth.state_dict(kind, seed=30, sharp=sharp, bias_jitter=0.0)
            fl = synth.film_params(1, seed=3)[0]
            m = ref_model(kind, nm, pm, sd, fl)
            tr = synth.t_rand(rays_g.shape[0], nc, seed=321)
            t = trace_render(pr, rays_g, 0.5, 1.5, m, m, nc, nf, tr)
            # round 1 named its sharp FiLM fixtures without a suffix: the plain-initialisation twin is "_soft"
            save(f"render_f5_{kind}_{nc}_{nf}{sharp_tag(sharp, '_soft')}", rays=rays_g, t_rand=tr, near=0.5, far=1.5, film=fl,
                 digest=np.array(synth.digest(sd)), **t)

        # F9: render_video (-> render_image per pose -> render_rays per 16 384-ray chunk; one torch.rand each)
        W, H, nc, nf = 12, 10, 16, 16
        poses = [nd.camera_pos_to_transform_matrix(4.0, th, -30.0) for th in (20.0, 200.0)]
        sd_c = synth.state_dict("nerf", seed=60, sharp="medium", bias_jitter=0.05)
        sd_f = synth.state_dict("nerf", seed=61, sharp="medium", bias_jitter=0.05)
        cm, fm = ref_model("nerf", nm, pm, sd_c), ref_model("nerf", nm, pm, sd_f)
        trs = [synth.t_rand(W * H, nc, seed=900 + i) for i in range(len(poses))]
        with injected_rand(list(trs)):
            rgb, depth, acc = nr.render_video(W, H, 1.3875 * W, poses, 2.0, 6.0, cm, fm, nc, nf)
        with injected_rand([trs[1]]):
            one = nr.render_image(W, H, 1.3875 * W, poses[1], 2.0, 6.0, cm, fm, nc, nf)
        with injected_rand([trs[0]]):
            one_np = pr.render_image_np(W, H, 1.3875 * W, poses[0], 2.0, 6.0, cm, fm, nc, nf)
        assert np.array_equal(one[0], rgb[1]) and np.array_equal(one_np[0], rgb[0])
        save("video_f9", W=W, H=H, focal=np.float64(1.3875 * W), near=2.0, far=6.0, n_coarse=nc, n_fine=nf,
             poses=np.stack(poses), t_rand=torch.stack(trs), rgb=rgb, depth=depth, acc=acc,
             digest_c=np.array(synth.digest(sd_c)), digest_f=np.array(synth.digest(sd_f)))

    # parameter layouts of the reference's own classes, for fields.detect_kind / pigan.Generator key parity
    gen = pm.Generator(256, 64)
    layouts = {
        "nerf.NeRF": nm.NeRF(), "nerf.SirenNeRF": nm.SirenNeRF(),
        "pi_GAN.FilmSirenNeRF(use_dir=True)": pm.FilmSirenNeRF(use_dir=True),
        "pi_GAN.FilmSirenNeRF(use_dir=False)": pm.FilmSirenNeRF(use_dir=False),
        "pi_GAN.Generator(256, 64)": gen, "pi_GAN.MappingNetwork()": pm.MappingNetwork(),
    }
    out = {name: {k: list(v.shape) for k, v in m.named_parameters()} for name, m in layouts.items()}
    out["pi_GAN.Generator(256, 64).state_dict"] = {k: list(v.shape) for k, v in gen.state_dict().items()}
    with open(os.path.join(HERE, "ref_layouts.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("ref_layouts.json", {k: len(v) for k, v in out.items()})


def make_r03(nm, pm):
    """ref_layer_attrs.json: what every layer object of the reference's field classes says about itself - class name,
    Dense.activation_name (nerf/nerf.py:15), FilmSiren.w_0 (pi_GAN/modules.py:15), and for a Linear inside a
    Sequential the class of the module after it (modules.py:81-84,89-92) - keyed like the state dict.  Data only;
    fields.hyper_mismatch is tested against it (tests/test_host_logic.py)."""
    import json
    models = {
        "nerf.NeRF": ("nerf", nm.NeRF()), "nerf.SirenNeRF": ("siren_nerf", nm.SirenNeRF()),
        "pi_GAN.FilmSirenNeRF(use_dir=True)": ("film_siren_nerf", pm.FilmSirenNeRF(use_dir=True)),
        "pi_GAN.FilmSirenNeRF(use_dir=False)": ("film_siren_nerf_nodir", pm.FilmSirenNeRF(use_dir=False)),
        "pi_GAN.FilmSirenNeRF(w_0=25)": ("film_siren_nerf", pm.FilmSirenNeRF(w_0=25)),
    }
    out = {}
    for name, (kind, m) in models.items():
        layers = {}
        for key, _ in SPECS[kind]:
            mod = m
            for part in key.split("."):
                mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
            rec = {"class": type(mod).__name__}
            for attr in ("activation_name", "w_0"):
                if hasattr(mod, attr):
                    rec[attr] = getattr(mod, attr)
            if key.endswith(".0") and isinstance(getattr(m, key[:-2], None), torch.nn.Sequential):
                rec["next_in_sequential"] = type(getattr(m, key[:-2])[1]).__name__
            layers[key] = rec
        out[name] = {"kind": kind, "layers": layers}
    with open(os.path.join(HERE, "ref_layer_attrs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("ref_layer_attrs.json", {k: len(v["layers"]) for k, v in out.items()})


FIT_REGIMES = [  # name, student kind, optimiser, lr, steps (every step takes all 3 456 training rays)
    ("fit_r03_siren_adam", "siren_nerf", "adam", 1e-5, 15), ("fit_r03_film_adam", "film_siren_nerf", "adam", 1e-5, 15),
    ("fit_r03_siren_sgd", "siren_nerf", "sgd", 2e-4, 8), ("fit_r03_film_sgd", "film_siren_nerf", "sgd", 2e-4, 8),
    ("fit_r03_siren_chaotic", "siren_nerf", "adam", 5e-4, 30),      # 256-ray batches: round 2's regime, diagnostic only
]


# Round 4: the HEADLINE model class (nerf/nerf.py:52-94 NeRF: PE, skip-concat layer 5, linear layers_dir[0], the 128-wide dir
# layer - the class BENCH's config and nerf/configs/lego.json name) followed over optimiser steps by the reference's own
# code.  Adam at train_nerf.py:98's 5e-4 is quiet enough here for the 1 % / 0.05 dB gate: the reference's rerun from
# weights perturbed by 1e-6 moves the losses by 5e-4 relative and the held-out PSNR by 7e-4 dB (a ReLU net's perturbation
# noise is switch flips, not amplification); plain SGD at 0.2 (the loss falls 12x in 8 steps; perturbed: 2e-4 per loss).
FIT_REGIMES_R04 = [("fit_r04_nerf_adam", "nerf", "adam", 5e-4, 20), ("fit_r04_nerf_sgd", "nerf", "sgd", 0.2, 8)]


def make_r03_fit(nr, nm, pm, regimes=None, save_scene=True):
    """F10 fit_r03_*: the loop of nerf/train_nerf.py:124-176 run by the REFERENCE's own code on this container's CPU - its
    render_rays (nerf/render.py:106-147), its SirenNeRF / FilmSirenNeRF modules, torch.optim.Adam / SGD - on the synthetic
    teacher scene of oracle/fit_ref.py (rays, teacher images - stored as fit_r03_scene.npz, because another host's CPU renders
    a few of their rays differently -, initial weights, per-step jitter: inputs), so that the GPU
    tests compare the HIP path's training trajectory with the reference's without re-running a 1-2 minute CPU fit per
    regime on the GPU box.  Stored: the loss of every step, the held-out view and its PSNR, and - to document the
    regime's own noise - the PSNR of the same run from initial weights perturbed by 1e-6 relative.  The oracle's own loop
    (fit_ref.fit_cpu) must reproduce the trajectory (asserted here: 1e-4 relative, the two being different code paths
    through the same arithmetic)."""
    from oracle import fit_ref

    def run(scene, student, optimizer, lr0, steps, batch, init):
        film = scene.film
        models = [ref_model(student, nm, pm, {k: v.clone() for k, v in sd.items()}, film) for sd in init]
        params = [p for m in models for p in m.parameters()]
        opt = fit_ref.make_optimizer(params, optimizer, lr0)
        losses = []
        for step in range(steps):
            rays, rgb, tr = scene.batch(step, batch)
            with injected_rand([tr]):
                out = nr.render_rays(rays, fit_ref.NEAR, fit_ref.FAR, models[0], models[1], scene.nc, scene.nf)
            loss = torch.mean((out[3] - rgb) ** 2) + torch.mean((out[0] - rgb) ** 2)          # train_nerf.py:158-166
            opt.zero_grad()
            loss.backward()
            opt.step()
            for g in opt.param_groups:
                g["lr"] = fit_ref.lr_at(step + 1, lr0)                                        # :170-175
            losses.append(float(loss.detach()))
        with torch.no_grad(), injected_rand([scene.heldout_jitter()]):
            held = nr.render_rays(scene.rays[-1], fit_ref.NEAR, fit_ref.FAR, models[0], models[1], scene.nc, scene.nf)
        return np.array(losses), held[3].numpy(), oref.psnr(held[3].numpy(), scene.images[-1].numpy())

    if save_scene:
        save("fit_r03_scene", images=torch.stack(fit_ref.Scene().images))      # the teacher's seven views as rendered HERE
    scene_images = None
    if not save_scene:                       # later rounds fit the SAME pictures (the committed fixture), not a re-render
        with np.load(os.path.join(HERE, "fit_r03_scene.npz")) as f:
            scene_images = f["images"]
    for name, student, optimizer, lr0, steps in (FIT_REGIMES if regimes is None else regimes):
        batch = 256 if name.endswith("chaotic") else 0
        scene = fit_ref.Scene(student=student, images=scene_images)
        losses, held, psnr = run(scene, student, optimizer, lr0, steps, batch, scene.student_init)
        o_losses, o_psnr, _ = fit_ref.fit_cpu(scene, steps, batch, lr0=lr0, optimizer=optimizer)
        rel = float(np.abs(np.array(o_losses) - losses).max() / losses.min())
        if not name.endswith("chaotic"):
            assert rel <= 1e-4 and abs(o_psnr - psnr) <= 2e-3, (name, rel, o_psnr, psnr)
        rng = np.random.Generator(np.random.PCG64(9))
        pert = tuple({k: v * torch.from_numpy((1 + 1e-6 * rng.standard_normal(tuple(v.shape))).astype(np.float32)) for k, v in sd.items()}
                     for sd in scene.student_init)
        p_losses, _, p_psnr = run(scene, student, optimizer, lr0, steps, batch, pert)
        save(name, losses=losses, heldout_rgb=held, heldout_psnr=np.float64(psnr), lr0=np.float64(lr0), steps=steps, batch=batch,
             optimizer=np.array(optimizer), student=np.array(student), digest_c=np.array(synth.digest(scene.student_init[0])),
             digest_f=np.array(synth.digest(scene.student_init[1])), oracle_loop_max_rel_loss_diff=np.float64(rel),
             oracle_loop_psnr=np.float64(o_psnr), perturbed_1e6_psnr=np.float64(p_psnr),
             perturbed_1e6_max_rel_loss_diff=np.float64(np.abs(p_losses - losses).max() / losses.min()))
        print(name, "loss", losses[0], "->", losses[-1], "psnr", psnr, "| oracle loop", o_psnr, "rel", rel, "| perturbed 1e-6", p_psnr, flush=True)


def main():
    if "--only-metrics" in sys.argv:
        make_metrics()
        return
    torch.set_num_threads(8)
    nr, nm, nd, pr, pm = load_reference()
    if "--only-r03" in sys.argv:
        make_r03(nm, pm)
        return
    if "--only-r03-fit" in sys.argv:
        make_r03_fit(nr, nm, pm)
        return
    if "--only-r04-fit" in sys.argv:
        make_r03_fit(nr, nm, pm, regimes=FIT_REGIMES_R04, save_scene=False)
        return
    if "--only-r02" in sys.argv:
        make_r02(nr, nm, nd, pr, pm)
        return

    # ---------------- F1: rays + poses ----------------
    pose_n = nd.camera_pos_to_transform_matrix(4.0, 37.0, -30.0)           # degrees
    pose_p = pr.camera_pos_to_transform_matrix(1.0, 0.2, -0.15)            # radians
    W, H, focal = 8, 6, 1.3875 * 8
    o, d = nr.get_rays(W, H, focal, pose_n)
    o2, d2 = pr.get_rays(W, H, float(W / 2 / np.tan(12 / 2 * np.pi / 180)), pose_p)
    save("rays_f1", W=W, H=H, focal=np.float64(focal), pose_nerf=pose_n, pose_pigan=pose_p,
         rays_o=np.ascontiguousarray(o), rays_d=d, rays_o_pigan=np.ascontiguousarray(o2), rays_d_pigan=d2,
         focal_pigan=np.float64(W / 2 / np.tan(12 / 2 * np.pi / 180)))

    # ---------------- F2: composite edge cases ----------------
    rng = np.random.Generator(np.random.PCG64(2))
    n, s = 12, 64
    raw = rng.uniform(0, 1, size=(n, s, 4)).astype(np.float32)
    raw[..., 3] *= 20.0
    z = np.sort(rng.uniform(2, 6, size=(n, s)).astype(np.float32), -1)
    raw[0, :, 3] = 0.0                 # empty ray
    raw[1, 0, 3] = 1e6                 # opaque first sample
    z[2, 10:14] = z[2, 10]             # equal depths
    raw[3, :, 3] = 1e-3                # faint everywhere
    raw[4, -1, 3] = 50.0               # mass on the last (delta = 1e10) sample
    raw[4, :-1, 3] = 0.0
    rd = rng.normal(size=(n, 3)).astype(np.float32) * 1.7
    outs = nr.raw_to_outputs(torch.from_numpy(raw), torch.from_numpy(z), torch.from_numpy(rd))
    save("composite_f2", raw=raw, z=z, rays_d=rd, rgb=outs[0], depth=outs[1], acc=outs[2], weights=outs[3])
    for s2 in (36, 192):
        raw2 = rng.uniform(0, 1, size=(40, s2, 4)).astype(np.float32)
        raw2[..., 3] = rng.exponential(3.0, size=(40, s2)).astype(np.float32) * (rng.random((40, s2)) < 0.3)
        z2 = np.sort(rng.uniform(0.5, 1.5, size=(40, s2)).astype(np.float32), -1)
        rd2 = rng.normal(size=(40, 3)).astype(np.float32)
        o_ = nr.raw_to_outputs(torch.from_numpy(raw2), torch.from_numpy(z2), torch.from_numpy(rd2))
        save(f"composite_f2_s{s2}", raw=raw2, z=z2, rays_d=rd2, rgb=o_[0], depth=o_[1], acc=o_[2], weights=o_[3])

    # ---------------- F3: sample_pdf edge cases ----------------
    nb = 63
    bins = np.sort(rng.uniform(2, 6, size=(10, nb)).astype(np.float32), -1)
    w = rng.uniform(0, 1, size=(10, nb - 1)).astype(np.float32) ** 4
    w[0] = 0.0                          # all-zero weights
    w[1] = 0.0; w[1, 17] = 1.0          # single spike
    w[2] = 0.0; w[2, 0] = 0.7; w[2, -1] = 0.3
    w[3] *= 1e-4                        # tiny mass
    w[4] = 1.0 / (nb - 1)
    pdf_arrays = dict(bins=bins, weights=w)
    for nf in (0, 1, 24, 128):
        zs = nr.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), nf)
        pdf_arrays[f"samples_{nf}"] = zs
    bins11 = np.sort(rng.uniform(0.5, 1.5, size=(10, 11)).astype(np.float32), -1)
    w11 = rng.uniform(0, 1, size=(10, 10)).astype(np.float32) ** 3
    pdf_arrays.update(bins11=bins11, weights11=w11,
                      samples11_24=nr.sample_pdf(torch.from_numpy(bins11), torch.from_numpy(w11), 24))
    save("pdf_f3", **pdf_arrays)

    # ---------------- F4: field MLPs ----------------
    x = sample_points(257, seed=4)
    film = synth.film_params(2, seed=1)
    f4 = dict(x=x, film=film)
    with torch.no_grad():
        for kind in ("nerf", "siren_nerf", "film_siren_nerf", "film_siren_nerf_nodir"):
            for sharp in (False, True):
                sd = synth.state_dict(kind, seed=10, sharp=sharp, bias_jitter=0.05)
                m = ref_model(kind, nm, pm, sd, film[1] if kind.startswith("film") else None)
                tag = f"{kind}{'_sharp' if sharp else ''}"
                f4[f"out.{tag}"] = m(x)
                f4[f"digest.{tag}"] = np.array(synth.digest(sd))
    save("field_f4", **f4)

    # ---------------- F5: render_rays with all intermediates ----------------
    def pick_rays(W_, H_, focal_, pose, n_rays, seed):
        return pick_rays_from(nr, W_, H_, focal_, pose, n_rays, seed)

    with torch.no_grad():
        pose = nd.camera_pos_to_transform_matrix(4.0, 63.0, -30.0)
        for kind, nc, nf, nrays, sharp in (
            ("nerf", 32, 0, 96, True), ("nerf", 64, 0, 96, True), ("nerf", 64, 128, 96, True),
            ("nerf", 64, 128, 64, False), ("siren_nerf", 64, 128, 64, False),
        ):
            rays = pick_rays(100, 100, 1.3875 * 100, pose, nrays, seed=nc + nf)
            sd_c = synth.state_dict(kind, seed=20, sharp=sharp, bias_jitter=0.05)
            sd_f = synth.state_dict(kind, seed=21, sharp=sharp, bias_jitter=0.05)
            cm, fm = ref_model(kind, nm, pm, sd_c), ref_model(kind, nm, pm, sd_f)
            tr = synth.t_rand(nrays, nc, seed=123)
            t = trace_render(nr, rays, 2.0, 6.0, cm, fm, nc, nf, tr)
            save(f"render_f5_{kind}_{nc}_{nf}{'_sharp' if sharp else ''}", rays=rays, t_rand=tr,
                 near=2.0, far=6.0, digest_c=np.array(synth.digest(sd_c)), digest_f=np.array(synth.digest(sd_f)), **t)
        # pi_GAN flavour: one FiLM field used for both passes, near/far 0.5/1.5, fov 12
        pose_g = pr.camera_pos_to_transform_matrix(1.0, 0.2, -0.1)
        focal_g = float(32 / 2 / np.tan(12 / 2 * np.pi / 180))
        o_, d_ = pr.get_rays(32, 32, focal_g, pose_g)
        rays_g = torch.from_numpy(np.stack([o_, d_], 2).reshape(-1, 2, 3).astype(np.float32)[::8].copy())
        for kind in ("film_siren_nerf", "film_siren_nerf_nodir"):
            sd = synth.state_dict(kind, seed=30, sharp=True, bias_jitter=0.0)
            fl = synth.film_params(1, seed=3)[0]
            m = ref_model(kind, nm, pm, sd, fl)
            tr = synth.t_rand(rays_g.shape[0], 12, seed=321)
            t = trace_render(pr, rays_g, 0.5, 1.5, m, m, 12, 24, tr)
            save(f"render_f5_{kind}_12_24", rays=rays_g, t_rand=tr, near=0.5, far=1.5, film=fl,
                 digest=np.array(synth.digest(sd)), **t)

    # ---------------- F6: pi_GAN image + grads ----------------
    res, b, nc, nf = 16, 2, 12, 24
    sd = synth.state_dict("film_siren_nerf", seed=40, sharp=True)
    gen = pm.Generator(64, res, near=0.5, far=1.5, fov=12, coarse_samples=nc, fine_samples=nf)
    gen.film_siren_nerf.load_state_dict(sd)
    film = synth.film_params(b, seed=5).clone().requires_grad_(True)
    thetas, phis = [0.2, -0.15], [0.05, -0.1]
    trs = [synth.t_rand(res * res, nc, seed=500 + i) for i in range(b)]
    imgs = []
    with injected_rand([t for t in trs]):
        for i in range(b):                              # Generator.forward loop, pi_GAN/modules.py:179-181
            gen.film_siren_nerf.set_film_params(film[i])
            imgs.append(gen.renderer(gen.film_siren_nerf, thetas[i], phis[i]))
    img = torch.stack(imgs)                             # [b,H,W,3]
    cot = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).normal(size=tuple(img.shape)).astype(np.float32))
    loss = (img * cot).sum()
    loss.backward()
    named = [(k, p.grad) for k, p in gen.film_siren_nerf.named_parameters()]
    save("pigan_grad_f6", res=res, near=0.5, far=1.5, fov=12.0, n_coarse=nc, n_fine=nf, thetas=np.array(thetas),
         phis=np.array(phis), film=film, t_rand=torch.stack(trs), image=img, cotangent=cot, loss=loss,
         grad_film=film.grad, digest=np.array(synth.digest(sd)), **grad_summary(named))

    # ---------------- F7: nerf training-loss grads ----------------
    nc, nf, nrays = 64, 128, 48
    pose = nd.camera_pos_to_transform_matrix(4.0, -120.0, -30.0)
    rays = pick_rays(100, 100, 1.3875 * 100, pose, nrays, seed=77)
    sd_c = synth.state_dict("nerf", seed=50, sharp=True, bias_jitter=0.05)
    sd_f = synth.state_dict("nerf", seed=51, sharp=True, bias_jitter=0.05)
    cm, fm = ref_model("nerf", nm, pm, sd_c), ref_model("nerf", nm, pm, sd_f)
    tr = synth.t_rand(nrays, nc, seed=9)
    tgt = torch.from_numpy(np.random.Generator(np.random.PCG64(8)).random((nrays, 4), dtype=np.float32))
    with injected_rand([tr]):
        rgb_c, _, acc_c, rgb_f, _, acc_f = nr.render_rays(rays, 2.0, 6.0, cm, fm, nc, nf)
    # nerf/train_nerf.py:158-167 with use_alpha and use_fine_model on
    loss_c = torch.mean((rgb_c - tgt[:, :3]) ** 2) + 0.1 * torch.mean((acc_c - tgt[:, 3]) ** 2)
    loss_f = torch.mean((rgb_f - tgt[:, :3]) ** 2) + 0.1 * torch.mean((acc_f - tgt[:, 3]) ** 2)
    loss = loss_f + loss_c
    loss.backward()
    named = [("coarse." + k, p.grad) for k, p in cm.named_parameters()] + \
            [("fine." + k, p.grad) for k, p in fm.named_parameters()]
    save("nerf_grad_f7", rays=rays, t_rand=tr, target=tgt, near=2.0, far=6.0, n_coarse=nc, n_fine=nf,
         loss=loss, rgb_c=rgb_c, acc_c=acc_c, rgb_f=rgb_f, acc_f=acc_f,
         digest_c=np.array(synth.digest(sd_c)), digest_f=np.array(synth.digest(sd_f)), **grad_summary(named))
    make_metrics()
    make_r02(nr, nm, nd, pr, pm)
    print("specs:", {k: len(v) for k, v in SPECS.items()})


if __name__ == "__main__":
    main()
