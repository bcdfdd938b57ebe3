"""CPU: the parity gates themselves (oracle/parity.py) - they accept the oracle, reject a systematic error in a
minority of bins / a biased MLP, and pick the bound they report."""
import numpy as np
import pytest
import torch

from oracle import fields as ofields, parity, render_ref as R, synth


def _case(sharp, n=24, nc=16, nf=24):
    sd_c = synth.state_dict("tiny_nerf", seed=1, sharp=sharp, bias_jitter=0.05)
    sd_f = synth.state_dict("tiny_nerf", seed=2, sharp=sharp, bias_jitter=0.05)
    rays = torch.from_numpy(R.rays_from_camera(20, 20, 27.0, synth.pose_degrees(4.0, 10.0, -30.0))[150:150 + n])
    tr = synth.t_rand(n, nc, seed=4)
    fc, ff = ofields.make_field("tiny_nerf", sd_c), ofields.make_field("tiny_nerf", sd_f)
    f64 = tuple(ofields.make_field("tiny_nerf", {k: v.double() for k, v in sd.items()}) for sd in (sd_c, sd_f))
    with torch.no_grad():
        ref = R.render_rays(rays, 2.0, 6.0, fc, ff, nc, nf, tr)
    return ref, f64, rays, tr, fc, ff, nc, nf


def test_gate_picks_and_reports_the_active_bound():
    ref = np.zeros(8)
    r = parity.gate("t", "s", "q", ref + 5e-5, ref)
    assert r["active"] == "hard" and r["passed"]
    with pytest.raises(AssertionError):
        parity.gate("t", "s", "q", ref + 2e-4, ref)                      # no fp64 evaluation: flat gate only
    ref64 = ref + 3e-4                                                    # the fp32 oracle sits 3e-4 from fp64
    r = parity.gate("t", "s", "q", ref + 2e-4, ref, ref64)
    assert r["active"] == "fp64-bound" and r["passed"] and r["err_vs_fp64"] < r["fp64_bound"]
    with pytest.raises(AssertionError):
        parity.gate("t", "s", "q", ref - 2e-4, ref, ref64)                # 5e-4 from fp64 > 1.5 * 3e-4 + 1e-5


@pytest.mark.parametrize("sharp", ["medium", True])
def test_check_render_accepts_the_oracle_and_rejects_errors(sharp):
    ref, f64, rays, tr, fc, ff, nc, nf = _case(sharp)
    chain = {k: getattr(ref, k).clone() for k in R.RenderTrace._fields}
    rec = parity.check_render("selfcheck", chain, ref, f64, rays, 2.0, 6.0, nc, nf, tr, fc, ff, sharp=sharp is True)
    assert rec["passed"] and rec["frac_rays_over"] == 0.0
    # a systematic error in a minority of bins: every resampled depth of one ray moved by a third of a bin
    bad = {k: v.clone() for k, v in chain.items()}
    bad["z_samples"][3] += 0.03
    bad["z_fine"] = torch.sort(torch.cat([ref.z_coarse, bad["z_samples"]], -1), -1).values
    with pytest.raises(AssertionError):
        parity.check_render("selfcheck-bad-bins", bad, ref, f64, rays, 2.0, 6.0, nc, nf, tr, fc, ff, sharp=sharp is True)
    # a biased fine pass on one ray
    bad = {k: v.clone() for k, v in chain.items()}
    bad["rgb_f"][5] += 1e-3
    with pytest.raises(AssertionError):
        parity.check_render("selfcheck-bad-rgb", bad, ref, f64, rays, 2.0, 6.0, nc, nf, tr, fc, ff, sharp=sharp is True)
    del parity.RECORDS[:]                                                # keep the CPU session's record file empty


def test_gate_grad_sees_magnitude_and_single_unit_errors():
    """The gradient gate (oracle/parity.py:gate_grad): accepts fp32 rounding, rejects a gradient that is 0.2 % too
    large everywhere (an L2 error), and - with an element gate - a derivative that is wrong on ONE unit of 65 536 (which
    moves the tensor's L2 error by 4e-5, far inside the norm gate: ADVICE r02 on the rebuilt sin derivative)."""
    rng = np.random.Generator(np.random.PCG64(1))
    r64 = rng.standard_normal(65536)
    r32 = r64.astype(np.float32)
    ok = parity.gate_grad("t", "w", r32 * np.float32(1 + 1e-6), r32, r64, elem_tol=parity.GRAD_ELEM_TOL_SMOOTH)
    assert ok["passed"] and ok["active"] == "hard" and ok["rel_l2_err"] < 2e-6
    with pytest.raises(AssertionError):
        parity.gate_grad("t", "w", r32 * np.float32(1.002), r32, r64)
    one = r32.copy()
    one[1234] += 0.01                                  # 1 % of the RMS on a single element
    assert parity.gate_grad("t", "w", one, r32, r64)["passed"]            # the norm gate alone does not see it
    with pytest.raises(AssertionError):
        parity.gate_grad("t", "w", one, r32, r64, elem_tol=parity.GRAD_ELEM_TOL_SMOOTH)
    # ReLU-style: the fp32 oracle itself sits 2e-3 from fp64 (a few derivative flips) - the HIP path may be 3x that
    flips = r64.copy()
    flips[:40] = 0.0
    noisy32 = flips.astype(np.float32)
    other = r64.copy()
    other[100:150] = 0.0
    rec = parity.gate_grad("t", "w", other.astype(np.float32), noisy32, r64, tol=1e-4)
    assert rec["active"] == "fp64-bound" and rec["passed"]
    del parity.RECORDS[:]
