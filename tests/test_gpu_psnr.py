"""GPU: the north star's "PSNR within 0.05 dB of the reference" on TRAINED results, without a dataset.

A student pair (coarse + fine) is fitted to a synthetic teacher-field scene (oracle/fit_ref.py: six 24x24 views of a
fixed seeded field rendered by the oracle, one view held out) with the loop of nerf/train_nerf.py:124-176 on the HIP path
(render_rays with autograd + a torch optimiser on the device) and compared with the SAME loop on the CPU - same initial
weights, same rays, same injected jitter.  Hard gates, every regime: the two loss curves within the regime's gate at
EVERY step and the two held-out-view PSNRs within 0.05 dB.

The CPU side of the sin-family regimes is the REFERENCE's own code - its render_rays, its SirenNeRF / FilmSirenNeRF
modules, torch's optimisers - run in the build container by tests/golden/make_golden.py --only-r03-fit and committed
as fixtures F10 (fit_r03_*.npz: every step's loss, the held-out view, its PSNR; a CPU fit of an 8x256 sin pair over all
3 456 rays costs 1-2 minutes per regime on the GPU box's host, which the suite cannot afford five times); the
generator asserts that the oracle's loop (fit_ref.fit_cpu) reproduces each trajectory.  The TinyNeRF regime (a
build-defined class the reference does not have) runs the oracle's loop live and also renders the CPU-trained field
with the HIP path.

Regimes and their own noise (the fixtures carry it: the same reference run from initial weights perturbed by 1e-6
relative - about what separates two fp32 forward passes; a regime can only gate another fp32 implementation at 1 % /
0.05 dB if that moves it by far less; tools/probes/fit_regimes.py, profiles/r03_fit_regimes.log):

  tiny_nerf        Adam 5e-4, 256-ray batches, 60 steps     PE + ReLU family          fp32 vs fp64: 1e-3 / 6e-4 dB
  siren_nerf       Adam 1e-5, all 3 456 rays, 15 steps      eight sin(30 u) layers    perturbed: 5e-5 / 2e-4 dB
  film_siren_nerf  Adam 1e-5, all rays, 15 steps, fixed FiLM row (the field alone trains, as in pi_GAN/synthesis.py:83-107)
                                                                                      perturbed: 1e-4 / 4e-4 dB
  nerf             (round 4, fixtures fit_r04_nerf_*) Adam 5e-4 - train_nerf.py:98's own rate -, all rays, 20 steps; plain SGD
                   0.2, 8 steps: the reference's NeRF class (nerf/nerf.py:52-94)   perturbed: 5e-4 / 7e-4 dB; 2e-4 per loss
  siren_nerf / film_siren_nerf   plain SGD 2e-4, all rays, 8 steps: Adam's first updates are m/sqrt(v) = +-1 per element
                   whatever the gradient's size, so only a step PROPORTIONAL to the gradient shows a gradient
                   of the wrong magnitude in the next loss; the loss falls 6x / 18x in these 8 steps.  Perturbed by 1e-6 the
                   SirenNeRF run moves by 4e-6 per step (gated at 1e-4), the FiLM run - whose first step takes the loss from 0.286
                   to 0.211 - by 2.5e-4 (gated at 2e-3)

  Adam at pi_GAN/train.py's 5e-5 is NOT such a regime: the loss rises 50 % over its first two steps, and that transient
  turns a 1e-7 perturbation into 6e-3 dB - fp32 vs fp64 on one host agreed to 1e-4 dB there, yet two hosts' CPUs landed
  0.12 dB apart with the HIP path in between (this round's first attempt).  Quiet under a change of precision is not
  quiet under perturbation.

Round 2's SirenNeRF regime (Adam 5e-4, 256-ray batches, 30 steps) is chaotic - the CPU loop in fp32 and in fp64 end
12-17 % / 0.02-0.5 dB apart - and stays as a recorded DIAGNOSTIC only (its first two steps, before any drift, are still
gated at 1e-5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fit_ref, parity, render_ref as R, synth  # noqa: E402


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
    rgb_held = held[3].cpu().numpy()
    return losses, R.psnr(rgb_held, scene.images[-1].numpy()), rgb_held


def test_tiny_nerf_fit_matches_the_cpu_loop_run_live():
    from mirender import fields, render_core
    steps, batch = 60, 256
    scene = fit_ref.Scene(student="tiny_nerf")
    cpu_losses, cpu_psnr, (sd_c, sd_f) = fit_ref.fit_cpu(scene, steps, batch)
    hip_losses, hip_psnr, _ = fit_hip(scene, steps, batch)
    rel = np.abs(np.array(hip_losses) - np.array(cpu_losses)) / np.array(cpu_losses)
    # the CPU-trained weights rendered by the HIP path: the renderer alone, on a trained field
    cm2, fm2 = fields.field_from_state_dict(sd_c, dev()), fields.field_from_state_dict(sd_f, dev())
    with torch.no_grad():
        held = render_core.render_rays(scene.rays[-1].to(dev()), fit_ref.NEAR, fit_ref.FAR, cm2, fm2, scene.nc, scene.nf,
                                       t_rand=scene.heldout_jitter().to(dev()))
    cross_psnr = R.psnr(held[3].cpu().numpy(), scene.images[-1].numpy())
    ok = bool(cpu_losses[-1] < 0.1 * cpu_losses[0] and rel.max() <= 0.01 and abs(hip_psnr - cpu_psnr) <= 0.05 and abs(cross_psnr - cpu_psnr) <= 0.05)
    parity.record(case=f"teacher scene fit (tiny_nerf) 24x24 16+16, {steps} adam steps (lr 0.0005) of {batch} rays",
                  stage="training trajectory", qty="held-out PSNR (dB)", hip=hip_psnr, cpu_reference_loop=cpu_psnr,
                  cpu_side="oracle loop (fit_ref.fit_cpu) run live on this host", cpu_weights_rendered_by_hip=cross_psnr,
                  err_vs_oracle32=abs(hip_psnr - cpu_psnr), tol=0.05, max_rel_loss_diff=float(rel.max()), rel_loss_gate=0.01,
                  final_loss_hip=hip_losses[-1], final_loss_cpu=cpu_losses[-1], active="hard", passed=ok)
    assert cpu_losses[-1] < 0.1 * cpu_losses[0]                            # the fit really trains
    assert rel.max() <= 0.01, (int(rel.argmax()), float(rel.max()))
    assert abs(hip_psnr - cpu_psnr) <= 0.05, (hip_psnr, cpu_psnr)
    assert abs(cross_psnr - cpu_psnr) <= 0.05, (cross_psnr, cpu_psnr)


# fixture, loss gate (relative, every step), the fit must reach this fraction of its first loss
REFERENCE_RUNS = [("fit_r03_siren_adam", 0.01, 0.2), ("fit_r03_film_adam", 0.01, 0.2),
                  ("fit_r03_siren_sgd", 1e-4, 0.3), ("fit_r03_film_sgd", 2e-3, 0.3),
                  # round 4: the headline class, nerf/nerf.py:52-94 NeRF (PE, skip-concat layer 5, linear layers_dir[0], 128-wide
                  # dir layer), Adam at train_nerf.py:98's 5e-4 over 20 steps and plain SGD 0.2 over 8 (the loss falls 22x / 12x).
                  # The reference's rerun from weights perturbed by 1e-6: 5e-4 / 2e-4 relative per loss, 7e-4 / 5e-5 dB
                  ("fit_r04_nerf_adam", 0.01, 0.2), ("fit_r04_nerf_sgd", 2e-3, 0.3)]


@pytest.mark.parametrize("name,rel_gate,must_reach", REFERENCE_RUNS)
def test_fit_matches_the_reference_codes_own_trajectory(golden, name, rel_gate, must_reach):
    g = golden(name)
    student, optimizer, lr0, steps, batch = str(g["student"]), str(g["optimizer"]), float(g["lr0"]), int(g["steps"]), int(g["batch"])
    scene = fit_ref.Scene(student=student, images=golden("fit_r03_scene")["images"])     # the pictures the reference run fitted
    assert synth.digest(scene.student_init[0]) == str(g["digest_c"]) and synth.digest(scene.student_init[1]) == str(g["digest_f"])
    ref_losses, ref_psnr = g["losses"], float(g["heldout_psnr"])
    hip_losses, hip_psnr, held = fit_hip(scene, steps, batch, lr0, optimizer)
    rel = np.abs(np.array(hip_losses) - ref_losses) / ref_losses
    held_mse = float(np.mean((held.astype(np.float64) - g["heldout_rgb"]) ** 2))
    trained = ref_losses[-1] < must_reach * ref_losses[0]
    ok = bool(trained and rel.max() <= rel_gate and abs(hip_psnr - ref_psnr) <= 0.05)
    parity.record(case=f"teacher scene fit ({student}) 24x24 16+16, {steps} {optimizer} steps (lr {lr0:g}) of all 3456 rays",
                  stage="training trajectory", qty="held-out PSNR (dB)", hip=hip_psnr, cpu_reference_loop=ref_psnr,
                  cpu_side=f"the reference's own render_rays / modules / torch optimiser, build container (tests/golden/{name}.npz)",
                  err_vs_oracle32=abs(hip_psnr - ref_psnr), tol=0.05, max_rel_loss_diff=float(rel.max()), rel_loss_gate=rel_gate,
                  rel_loss_diff_per_step=[float(x) for x in rel], first_loss=float(ref_losses[0]), final_loss_hip=hip_losses[-1],
                  final_loss_cpu=float(ref_losses[-1]), heldout_view_psnr_hip_vs_reference_image=-10 * np.log10(max(held_mse, 1e-30)),
                  regime_noise_reference_rerun_from_weights_perturbed_1e6=dict(
                      psnr_db=abs(float(g["perturbed_1e6_psnr"]) - ref_psnr), max_rel_loss=float(g["perturbed_1e6_max_rel_loss_diff"])),
                  oracle_loop_vs_reference=dict(psnr_db=abs(float(g["oracle_loop_psnr"]) - ref_psnr),
                                                max_rel_loss=float(g["oracle_loop_max_rel_loss_diff"])),
                  active="hard", passed=ok)
    assert trained, (ref_losses[0], ref_losses[-1])                         # the fit really trains
    assert rel.max() <= rel_gate, (int(rel.argmax()), float(rel.max()), rel_gate)
    assert abs(hip_psnr - ref_psnr) <= 0.05, (hip_psnr, ref_psnr)


def test_chaotic_siren_regime_is_recorded_as_a_diagnostic(golden):
    """Round 2's SirenNeRF regime (Adam 5e-4, 256-ray batches, 30 steps): two correct implementations of this loop end
    10-20 % apart in the loss (the CPU loop in fp32 vs fp64: 12 % / 0.28 dB), so its curve gates nothing - it is
    recorded against the reference's own run (fixture).  What IS deterministic in it is gated: the first two steps
    (forward, first gradients' signs, first Adam update) at 1e-5."""
    g = golden("fit_r03_siren_chaotic")
    steps, batch = int(g["steps"]), int(g["batch"])
    scene = fit_ref.Scene(student="siren_nerf", images=golden("fit_r03_scene")["images"])
    hip_losses, hip_psnr, _ = fit_hip(scene, steps, batch)
    rel = np.abs(np.array(hip_losses) - g["losses"]) / g["losses"]
    parity.record(case=f"teacher scene fit (siren_nerf, CHAOTIC regime) 24x24 16+16, {steps} adam steps (lr 5e-4) of {batch} rays",
                  stage="training trajectory (diagnostic)", qty="held-out PSNR (dB)", hip=hip_psnr,
                  cpu_reference_loop=float(g["heldout_psnr"]), err_vs_oracle32=abs(hip_psnr - float(g["heldout_psnr"])), tol=None,
                  max_rel_loss_diff=float(rel.max()), rel_loss_diff_first_two_steps=float(rel[:2].max()),
                  regime_noise_reference_rerun_from_weights_perturbed_1e6=dict(
                      psnr_db=abs(float(g["perturbed_1e6_psnr"]) - float(g["heldout_psnr"])),
                      max_rel_loss=float(g["perturbed_1e6_max_rel_loss_diff"])),
                  active="diagnostic", passed=bool(rel[:2].max() <= 1e-5))
    assert rel[:2].max() <= 1e-5, rel[:4]
    assert np.isfinite(hip_losses).all() and hip_losses[-1] < hip_losses[0]
