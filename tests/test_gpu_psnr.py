"""GPU: the north star's "PSNR within 0.05 dB of the reference" on a trained result, without a dataset.

A TinyNeRF pair (coarse + fine) is fitted to a synthetic teacher-field scene (oracle/fit_ref.py: six 24x24 views of a
fixed seeded field rendered by the oracle, one view held out) with the loop of nerf/train_nerf.py:124-176, twice: on
the HIP path (render_rays with autograd + torch Adam on the device) and on the CPU by autograd through the oracle -
same initial weights, same ray batches, same injected jitter.  The two loss curves must agree within 1 % at every
step and the two held-out-view PSNRs within 0.05 dB (TinyNeRF).  A SirenNeRF pair runs the same loop; its training is
chaotic at this learning rate (the CPU loop in fp32 and in fp64 drift 12-17 % / 0.02-0.5 dB apart depending on the
host), so its hard gates are the first two steps at 1e-5 (forward, first gradients, first Adam update) and the renderer
alone - the CPU-trained field rendered by the HIP path - at 0.05 dB; the rest of the curve is recorded next to the
CPU's own fp32 / fp64 drift and sanity-gated."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fit_ref, parity, render_ref as R  # noqa: E402

STEPS, BATCH = 60, 256
SIREN_STEPS = 30     # the siren loop costs three CPU fits of an 8 x 256 pair (fp32, fp64, and the HIP run's host side)


def dev():
    return torch.device("cuda", 0)


def fit_hip(scene, steps, batch_size):
    from mirender import fields, render_core
    cm = fields.field_from_state_dict(scene.student_init[0], dev())
    fm = fields.field_from_state_dict(scene.student_init[1], dev())
    opt = torch.optim.Adam(list(cm.parameters()) + list(fm.parameters()), lr=5e-4, betas=(0.9, 0.999))
    losses = []
    for step in range(steps):
        rays, rgb, tr = scene.batch(step, batch_size)
        rgb = rgb.to(dev())
        out = render_core.render_rays(rays.to(dev()), fit_ref.NEAR, fit_ref.FAR, cm, fm, scene.nc, scene.nf, t_rand=tr.to(dev()))
        loss = torch.mean((out[3] - rgb) ** 2) + torch.mean((out[0] - rgb) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g["lr"] = fit_ref.lr_at(step + 1)
        losses.append(float(loss.detach()))
    with torch.no_grad():
        held = render_core.render_rays(scene.rays[-1].to(dev()), fit_ref.NEAR, fit_ref.FAR, cm, fm, scene.nc, scene.nf,
                                       t_rand=scene.heldout_jitter().to(dev()))
    return losses, R.psnr(held[3].cpu().numpy(), scene.images[-1].numpy()), (cm, fm)


@pytest.mark.parametrize("student", ["tiny_nerf", "siren_nerf"])
def test_fit_to_teacher_scene_matches_the_cpu_reference_loop(student):
    """tiny_nerf: positional encoding + ReLU (the nerf family); siren_nerf: eight sin(30 u) layers, whose training
    amplifies any difference in the activation's arithmetic."""
    from mirender import render_core
    scene = fit_ref.Scene(student=student)
    steps = SIREN_STEPS if student == "siren_nerf" else STEPS
    cpu_losses, cpu_psnr, (sd_c, sd_f) = fit_ref.fit_cpu(scene, steps, BATCH)
    hip_losses, hip_psnr, (cm, fm) = fit_hip(scene, steps, BATCH)
    rel = np.abs(np.array(hip_losses) - np.array(cpu_losses)) / np.array(cpu_losses)
    assert cpu_losses[-1] < (0.1 if student == "tiny_nerf" else 0.7) * cpu_losses[0]      # the fit really trains
    # the CPU-trained weights rendered by the HIP path: the renderer alone, on a trained field
    from mirender import fields
    cm2, fm2 = fields.field_from_state_dict(sd_c, dev()), fields.field_from_state_dict(sd_f, dev())
    with torch.no_grad():
        held = render_core.render_rays(scene.rays[-1].to(dev()), fit_ref.NEAR, fit_ref.FAR, cm2, fm2, scene.nc, scene.nf,
                                       t_rand=scene.heldout_jitter().to(dev()))
    cross_psnr = R.psnr(held[3].cpu().numpy(), scene.images[-1].numpy())
    # How far two correct fp32 implementations of this loop may drift: the same CPU loop in double.  tiny_nerf: 1e-3 in
    # the losses, 6e-4 dB; siren_nerf: the sin(30 u) stack under Adam's first normalised steps amplifies rounding
    # differences 10-100x per step - the CPU loop in fp32 and in fp64 are 17 % apart in the losses and 0.5 dB in PSNR
    # after 60 steps, 12 % after 30 (and two hosts' MKL builds 0.2 dB), so the trajectory gates for it are stated against
    # that drift.
    drift = dict(max_rel=0.0, psnr=0.0)
    if student != "tiny_nerf":
        l64, p64, _ = fit_ref.fit_cpu(scene, steps, BATCH, f64=True)
        drift = dict(max_rel=float((np.abs(np.array(cpu_losses) - np.array(l64)) / np.array(l64)).max()), psnr=abs(cpu_psnr - p64))
    # siren_nerf: where the curve lands after its chaotic steps is recorded next to the CPU's own fp32 / fp64 drift and
    # only sanity-gated (a wrong gradient shows at step 1, which IS gated hard below: 1e-5); which of two close
    # trajectories a host's MKL build follows differs from CPU model to CPU model
    rel_gate = 0.01 if student == "tiny_nerf" else max(5.0 * drift["max_rel"], 0.5)
    psnr_gate = 0.05 if student == "tiny_nerf" else max(1.0, 3.0 * drift["psnr"])
    first_gate = 0.01 if student == "tiny_nerf" else 1e-5       # smooth activations: the first two steps agree to rounding
    ok = bool(abs(hip_psnr - cpu_psnr) <= psnr_gate and rel.max() <= rel_gate and rel[:2].max() <= first_gate and abs(cross_psnr - cpu_psnr) <= 0.05)
    parity.record(case=f"teacher scene fit ({student}) 24x24 16+16, {steps} Adam steps of {BATCH} rays", stage="training trajectory",
                  qty="held-out PSNR (dB)", hip=hip_psnr, cpu_reference_loop=cpu_psnr, cpu_weights_rendered_by_hip=cross_psnr,
                  err_vs_oracle32=abs(hip_psnr - cpu_psnr), tol=psnr_gate, max_rel_loss_diff=float(rel.max()), rel_loss_gate=rel_gate,
                  rel_loss_diff_first_two_steps=float(rel[:2].max()), cpu_fp32_vs_fp64_max_rel_loss=drift["max_rel"],
                  cpu_fp32_vs_fp64_psnr_db=drift["psnr"], final_loss_hip=hip_losses[-1], final_loss_cpu=cpu_losses[-1],
                  active="hard" if student == "tiny_nerf" else "cpu-drift", passed=ok)
    assert rel[:2].max() <= first_gate, rel[:4]                      # before any drift: forward and first gradients agree
    assert rel.max() <= rel_gate, (int(rel.argmax()), float(rel.max()), rel_gate)
    assert abs(hip_psnr - cpu_psnr) <= psnr_gate, (hip_psnr, cpu_psnr, psnr_gate)
    assert abs(cross_psnr - cpu_psnr) <= 0.05, (cross_psnr, cpu_psnr)     # the renderer alone on the CPU-trained field
