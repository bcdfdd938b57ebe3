"""GPU: the north star's "PSNR within 0.05 dB of the reference" on TRAINED results, without a dataset.

A student pair (coarse + fine) is fitted to a synthetic teacher-field scene (oracle/fit_ref.py: six 24x24 views of a
fixed seeded field rendered by the oracle, one view held out) with the loop of nerf/train_nerf.py:124-176, twice: on
the HIP path (render_rays with autograd + a torch optimiser on the device) and on the CPU by autograd through the
oracle - same initial weights, same rays, same injected jitter.  Hard gates, every regime: the two loss curves within
1 % at EVERY step and the two held-out-view PSNRs within 0.05 dB.

Regimes (tools/probes/fit_regimes.py measured, in the build container, how far the CPU loop in fp32 drifts from the
same loop in fp64 in each - a regime can only gate another fp32 implementation at 1 % / 0.05 dB if that drift is far
below the gate; profiles/r03_fit_regimes.log):

  tiny_nerf        Adam 5e-4, 256-ray batches, 60 steps     PE + ReLU family            drift 1e-3 / 6e-4 dB
  siren_nerf       Adam 1e-5, all 3 456 rays, 15 steps      eight sin(30 u) layers      1e-6 perturbation: 2e-4 dB
  film_siren_nerf  Adam 1e-5, all rays, 15 steps, fixed FiLM row (the field alone trains, as in
                   pi_GAN/synthesis.py:83-107)               FiLM sin layers             1e-6 perturbation: 4e-4 dB
                   (Adam at pi_GAN/train.py's 5e-5 overshoots - the loss RISES 50 % over its first two steps - and that
                   transient amplifies a 1e-7 relative perturbation of the initial weights into 6e-3 dB; fp32 vs fp64 on one
                   host looked quiet there (1e-4 dB) but two hosts' CPUs landed 0.12 dB apart, the HIP path in between: a
                   regime has to be quiet under PERTURBATION, not only under a change of precision, to carry a 0.05 dB gate)
  siren_nerf / film_siren_nerf   plain SGD 2e-4, all rays, 8 steps: Adam's first updates are m/sqrt(v) = +-1 per element
                   whatever the gradient's size, so only a step PROPORTIONAL to the gradient shows a gradient of
                   the wrong magnitude in the next loss; the loss falls 6x in these 8 steps   drift 3e-7: gated at 1e-4

The round-2 SirenNeRF regime (Adam 5e-4, 256-ray batches, 30 steps) is chaotic - the CPU loop in fp32 and in fp64 end
12-17 % / 0.02-0.5 dB apart - and stays as a recorded DIAGNOSTIC only (its first two steps, before any drift, are
still gated at 1e-5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fit_ref, parity, render_ref as R  # noqa: E402


def dev():
    return torch.device("cuda", 0)


def fit_hip(scene, steps, batch_size, lr0=5e-4, optimizer="adam"):
    from mirender import fields, render_core
    cm = fields.field_from_state_dict(scene.student_init[0], dev())
    fm = fields.field_from_state_dict(scene.student_init[1], dev())
    film = None if scene.film is None else scene.film.to(dev()).reshape(1, 9, 512)
    opt = fit_ref.make_optimizer(list(cm.parameters()) + list(fm.parameters()), optimizer, lr0)
    losses = []
    for step in range(steps):
        rays, rgb, tr = scene.batch(step, batch_size)
        rgb = rgb.to(dev())
        out = render_core.render_rays(rays.to(dev()), fit_ref.NEAR, fit_ref.FAR, cm, fm, scene.nc, scene.nf,
                                      t_rand=tr.to(dev()), film=film)
        loss = torch.mean((out[3] - rgb) ** 2) + torch.mean((out[0] - rgb) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g["lr"] = fit_ref.lr_at(step + 1, lr0)
        losses.append(float(loss.detach()))
    with torch.no_grad():
        held = render_core.render_rays(scene.rays[-1].to(dev()), fit_ref.NEAR, fit_ref.FAR, cm, fm, scene.nc, scene.nf,
                                       t_rand=scene.heldout_jitter().to(dev()), film=film)
    return losses, R.psnr(held[3].cpu().numpy(), scene.images[-1].numpy()), (cm, fm)


def render_cpu_weights_on_hip(scene, sd_c, sd_f):
    """The CPU-trained field rendered by the HIP path: the renderer alone, on a trained field."""
    from mirender import fields, render_core
    cm2, fm2 = fields.field_from_state_dict(sd_c, dev()), fields.field_from_state_dict(sd_f, dev())
    film = None if scene.film is None else scene.film.to(dev()).reshape(1, 9, 512)
    with torch.no_grad():
        held = render_core.render_rays(scene.rays[-1].to(dev()), fit_ref.NEAR, fit_ref.FAR, cm2, fm2, scene.nc, scene.nf,
                                       t_rand=scene.heldout_jitter().to(dev()), film=film)
    return R.psnr(held[3].cpu().numpy(), scene.images[-1].numpy())


# student, optimiser, lr, batch (0 = every training ray), steps, loss gate (relative, every step), PSNR gate (dB),
# the fit must reach this fraction of its first loss, the regime's own noise measured by tools/probes/fit_regimes.py
HARD = [("tiny_nerf", "adam", 5e-4, 256, 60, 0.01, 0.05, 0.1, "1e-3 rel loss, 6e-4 dB"),
        ("siren_nerf", "adam", 1e-5, 0, 15, 0.01, 0.05, 0.2, "weights perturbed by 1e-6 relative: 5e-5 rel loss, 2e-4 dB"),
        ("film_siren_nerf", "adam", 1e-5, 0, 15, 0.01, 0.05, 0.2, "weights perturbed by 1e-6 relative: 1e-4 rel loss, 4e-4 dB"),
        ("siren_nerf", "sgd", 2e-4, 0, 8, 1e-4, 0.05, 0.3, "fp32 vs fp64 3.0e-7 rel loss; weights perturbed by 1e-6: < 1e-4 dB"),
        ("film_siren_nerf", "sgd", 2e-4, 0, 8, 1e-4, 0.05, 0.3, "fp32 vs fp64 2.2e-7 rel loss; weights perturbed by 1e-6: < 1e-4 dB")]


@pytest.mark.parametrize("student,optimizer,lr0,batch,steps,rel_gate,psnr_gate,must_reach,drift", HARD)
def test_fit_to_teacher_scene_matches_the_cpu_reference_loop(student, optimizer, lr0, batch, steps, rel_gate, psnr_gate,
                                                             must_reach, drift):
    scene = fit_ref.Scene(student=student)
    cpu_losses, cpu_psnr, (sd_c, sd_f) = fit_ref.fit_cpu(scene, steps, batch, lr0=lr0, optimizer=optimizer)
    hip_losses, hip_psnr, _ = fit_hip(scene, steps, batch, lr0, optimizer)
    rel = np.abs(np.array(hip_losses) - np.array(cpu_losses)) / np.array(cpu_losses)
    cross_psnr = render_cpu_weights_on_hip(scene, sd_c, sd_f)
    trained = cpu_losses[-1] < must_reach * cpu_losses[0]
    ok = bool(trained and rel.max() <= rel_gate and abs(hip_psnr - cpu_psnr) <= psnr_gate and abs(cross_psnr - cpu_psnr) <= psnr_gate)
    parity.record(case=f"teacher scene fit ({student}) 24x24 16+16, {steps} {optimizer} steps (lr {lr0:g}) of "
                       f"{batch or 'all 3456'} rays", stage="training trajectory", qty="held-out PSNR (dB)", hip=hip_psnr,
                  cpu_reference_loop=cpu_psnr, cpu_weights_rendered_by_hip=cross_psnr, err_vs_oracle32=abs(hip_psnr - cpu_psnr),
                  tol=psnr_gate, max_rel_loss_diff=float(rel.max()), rel_loss_gate=rel_gate,
                  rel_loss_diff_first_two_steps=float(rel[:2].max()), first_loss=cpu_losses[0], final_loss_hip=hip_losses[-1],
                  final_loss_cpu=cpu_losses[-1], regime_noise_measured_in_build_container=drift,
                  active="hard", passed=ok)
    assert trained, (cpu_losses[0], cpu_losses[-1])                       # the fit really trains
    assert rel.max() <= rel_gate, (int(rel.argmax()), float(rel.max()), rel_gate)
    assert abs(hip_psnr - cpu_psnr) <= psnr_gate, (hip_psnr, cpu_psnr)
    assert abs(cross_psnr - cpu_psnr) <= psnr_gate, (cross_psnr, cpu_psnr)    # the renderer alone on the CPU-trained field


def test_chaotic_siren_regime_is_recorded_as_a_diagnostic():
    """Round 2's SirenNeRF regime (Adam 5e-4, 256-ray batches, 30 steps): two correct implementations of this loop end
    10-20 % apart in the loss (the CPU loop in fp32 vs fp64: 12 % / 0.28 dB), so its curve gates nothing - it is
    recorded.  What IS deterministic in it is gated: the first two steps (forward, first gradients' signs, first Adam
    update) at 1e-5, and the renderer alone on the CPU-trained field at 0.05 dB."""
    steps, batch = 30, 256
    scene = fit_ref.Scene(student="siren_nerf")
    cpu_losses, cpu_psnr, (sd_c, sd_f) = fit_ref.fit_cpu(scene, steps, batch)
    hip_losses, hip_psnr, _ = fit_hip(scene, steps, batch)
    rel = np.abs(np.array(hip_losses) - np.array(cpu_losses)) / np.array(cpu_losses)
    cross_psnr = render_cpu_weights_on_hip(scene, sd_c, sd_f)
    parity.record(case=f"teacher scene fit (siren_nerf, CHAOTIC regime) 24x24 16+16, {steps} adam steps (lr 5e-4) of {batch} rays",
                  stage="training trajectory (diagnostic)", qty="held-out PSNR (dB)", hip=hip_psnr, cpu_reference_loop=cpu_psnr,
                  cpu_weights_rendered_by_hip=cross_psnr, err_vs_oracle32=abs(hip_psnr - cpu_psnr), tol=None,
                  max_rel_loss_diff=float(rel.max()), rel_loss_diff_first_two_steps=float(rel[:2].max()),
                  cpu_fp32_vs_fp64_drift_measured_in_build_container="12 % rel loss, 0.28 dB after 30 steps",
                  active="diagnostic", passed=bool(rel[:2].max() <= 1e-5 and abs(cross_psnr - cpu_psnr) <= 0.05))
    assert rel[:2].max() <= 1e-5, rel[:4]
    assert abs(cross_psnr - cpu_psnr) <= 0.05, (cross_psnr, cpu_psnr)
    assert np.isfinite(hip_losses).all() and hip_losses[-1] < hip_losses[0]
