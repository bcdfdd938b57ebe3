"""Oracle (test infrastructure): the parity gates and the record every parity check leaves behind.

Only tests/ and __graft_entry__.smoke() import this.  Every check appends one record - case, stage, quantity,
achieved max |err| against the fp32 oracle (and against its fp64 evaluation where one was made), the gate and WHICH
bound was the active one - and the pytest session writes them to gpurun_out/r04_parity.json (committed copy of the round:
profiles/r04_parity.json).

Gates (BASELINE.json north_star: "within 1e-4 abs on fixed seeds"):

  hard        |HIP - oracle32| <= tol                (tol 1e-4; depth, which lives on a scale of 2..6, 5e-4)
  fp64-bound  only where a caller passes the fp64 evaluation AND the hard gate did not hold:
              |HIP - oracle64| <= 1.5 * |oracle32 - oracle64| + tol/10
              i.e. the HIP path may not sit further from exact arithmetic than the reference's own fp32 path does
              (synthetic "sharp" density heads, x50, push every fp32 pipeline past 1e-4).  Intermediates that carry
              no 1e-4 claim (sigma, compositing weights) use 2.0 instead of 1.5: see FP64_FACTOR_INTERMEDIATE.

The end-to-end fine pass is checked by composition instead of by a conditioning heuristic (`check_render`): the HIP
stages are chained through the C ABI, the chain must reproduce the fused render_rays call bit for bit, and every link
is gated hard against the oracle evaluated on THE SAME inputs (the HIP path's own coarse weights / fine depths), for
every ray.  The distance to the oracle's own end-to-end image is then reported as a distribution next to the fp32
oracle's distance from its fp64 self, and every ray over the gate must be a ray whose fine depths really differ.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import render_ref as R

TOL = 1e-4
DEPTH_TOL = 5e-4
RECORDS: list = []


def _np64(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def maxabs(a, b) -> float:
    a, b = _np64(a), _np64(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) if a.size else 0.0


def record(**kw):
    RECORDS.append({k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kw.items()})
    return RECORDS[-1]


FP64_FACTOR = 1.5            # outputs (rgb, acc, depth, alpha): HIP within 1.5x the fp32 oracle's distance from fp64
FP64_FACTOR_INTERMEDIATE = 2.0
# Intermediates with no 1e-4 claim of their own (unbounded sigma, compositing weights) get 2.0 on the max norm.
# Why not 1.5 everywhere: measured over 16 384 points (tools/probes/field_rms.py) the HIP kernels'
# RMS distance from fp64 equals that of torch/MKL on this build container's CPU (1.26e-7 vs 1.26e-7 on rgb: both
# are one fp32 FMA chain over k) while the GPU box's AVX-512 MKL kernel sits at 0.94e-7 - the signature of two
# interleaved accumulation chains (a 1.39x lower RMS in emulation).  With an RMS ratio of 1.35 between two fp32
# references, a max-norm ratio over a few thousand heavy-tailed samples lands anywhere in 1.0-1.7.


def gate(case: str, stage: str, qty: str, got, ref32, ref64=None, tol: float = TOL, check: bool = True,
         factor: float = FP64_FACTOR) -> dict:
    """One parity check (see the module docstring for the two bounds).  Returns its record."""
    e32 = maxabs(got, ref32)
    rec = dict(case=case, stage=stage, qty=qty, err_vs_oracle32=e32, tol=tol)
    ok, active = e32 <= tol, "hard"
    if ref64 is not None:
        e64, f = maxabs(got, ref64), maxabs(ref32, ref64)
        bound = factor * f + 0.1 * tol
        rec.update(err_vs_fp64=e64, oracle32_vs_fp64=f, fp64_bound=bound, fp64_factor=factor)
        if not ok:
            ok, active = e64 <= bound, "fp64-bound"
    rec.update(active=active, passed=bool(ok))
    record(**rec)
    if check:
        assert ok, rec
    return rec


GRAD_TOL_SMOOTH = 5e-4     # sin / FiLM networks: relative L2 error of a gradient tensor (norm of the error / norm of the tensor)
GRAD_TOL_RELU = 5e-3       # ReLU networks: a derivative is a 0/1 switch on a pre-activation's sign; two fp32 pipelines flip a
#                            handful of (point, unit) pairs and each flip moves a tensor's gradient by ~1e-3 of its norm
GRAD_ELEM_TOL_SMOOTH = 2e-3  # sin / FiLM networks: max over ELEMENTS of |error| / RMS(tensor) - sensitive to a wrong
#                              derivative factor on a few units (the rebuilt +-30 sqrt(1 - X^2)), which an L2 norm averages away


def gate_grad(case: str, tensor: str, got, ref32, ref64=None, tol: float = GRAD_TOL_SMOOTH, cpu_factor: float = 3.0,
              elem_tol: float | None = None, check: bool = True, stage: str = "gradient") -> dict:
    """One gradient tensor of the HIP backward path against the oracle's autograd (same cotangents, same inputs).

      rel_l2      ||got - ref|| / ||ref||, against the fp64 oracle where one is given (else the fp32 oracle)
      gate        rel_l2 <= tol ("hard"), else - only with ref64 - rel_l2 <= cpu_factor x the fp32 oracle's own rel_l2
                  from fp64 ("fp64-bound": the HIP path may not sit further from exact arithmetic than the reference's
                  fp32 autograd does, up to the factor)
      elem        max |got - ref| / RMS(ref): recorded always; gated (same two bounds, with elem_tol) when elem_tol is
                  given.  ADVICE r02: per-tensor norms alone would not see a derivative that is wrong on a few units.
    Leaves one record: both errors, the fp32 oracle's own, the gates and which bound was active."""
    g = _np64(got).reshape(-1)
    r32 = _np64(ref32).reshape(-1)
    ref = r32 if ref64 is None else _np64(ref64).reshape(-1)
    assert g.shape == ref.shape, (tensor, g.shape, ref.shape)
    scale = max(float(np.linalg.norm(ref)), 1e-30)
    rms = max(scale / np.sqrt(max(ref.size, 1)), 1e-30)
    e_hip, elem_hip = float(np.linalg.norm(g - ref)) / scale, float(np.abs(g - ref).max(initial=0.0)) / rms
    rec = dict(case=case, stage=stage, qty=tensor, rel_l2_err=e_hip, tol=tol, max_elem_err_over_rms=elem_hip,
               reference="oracle fp64 autograd" if ref64 is not None else "oracle fp32 autograd", elements=int(ref.size))
    ok, active = e_hip <= tol, "hard"
    if ref64 is not None:
        e_cpu, elem_cpu = float(np.linalg.norm(r32 - ref)) / scale, float(np.abs(r32 - ref).max(initial=0.0)) / rms
        rec.update(oracle32_rel_l2_from_fp64=e_cpu, oracle32_max_elem_over_rms=elem_cpu, cpu_factor=cpu_factor)
        if not ok:
            ok, active = e_hip <= cpu_factor * e_cpu, "fp64-bound"
    if elem_tol is not None:
        rec["elem_tol"] = elem_tol
        ok_e = elem_hip <= elem_tol
        if not ok_e and ref64 is not None:
            ok_e = elem_hip <= cpu_factor * rec["oracle32_max_elem_over_rms"]
            active = "fp64-bound" if ok_e else active
        ok = ok and ok_e
    rec.update(active=active, passed=bool(ok))
    record(**rec)
    if check:
        assert ok, rec
    return rec


def gate_grad_samples(case: str, tensor: str, got, idx, val, l2, tol: float, check: bool = True) -> dict:
    """A gradient tensor against a golden fixture that keeps 512 strided samples + the L2 norm per tensor
    (tests/golden/make_golden.py: the reference's own autograd).  Both in units of the tensor's norm."""
    g = _np64(got).reshape(-1)
    scale = max(float(l2), 1e-12)
    err = float(np.abs(g[np.asarray(idx)] - _np64(val)).max()) / scale
    nerr = abs(float(np.sqrt((g ** 2).sum())) - float(l2)) / scale
    rec = dict(case=case, stage="gradient (golden fixture)", qty=tensor, max_sample_err_over_norm=err,
               norm_err_over_norm=nerr, tol=tol, reference="reference autograd (fixture)", active="hard",
               passed=bool(err <= tol and nerr <= tol))
    record(**rec)
    if check:
        assert rec["passed"], rec
    return rec


def write_records(path: str):
    if not RECORDS:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    hard = [r for r in RECORDS if r.get("active") == "hard"]
    soft = [r for r in RECORDS if r.get("active") == "fp64-bound"]
    grads = [r for r in RECORDS if str(r.get("stage", "")).startswith("gradient")]
    summary = dict(checks=len(RECORDS), hard_gate_active=len(hard), fp64_bound_active=len(soft),
                   failed=sum(1 for r in RECORDS if r.get("passed") is False), gradient_records=len(grads),
                   worst_gradient_rel_l2=max((r["rel_l2_err"] for r in grads if "rel_l2_err" in r), default=0.0),
                   worst_hard=max((r["err_vs_oracle32"] / r["tol"] for r in hard if "err_vs_oracle32" in r and r.get("tol")),
                                  default=0.0),
                   worst_fp64_ratio=max((r["err_vs_fp64"] / max(r["oracle32_vs_fp64"], 1e-30) for r in soft
                                         if "err_vs_fp64" in r), default=0.0))
    with open(path, "w") as f:
        json.dump(dict(summary=summary, records=RECORDS), f, indent=1)


# ----------------------------------------------------------------------------------------------------------------
# inverse-CDF conditioning (render.py:27-56)
# ----------------------------------------------------------------------------------------------------------------
def pdf_conditioning(bins, weights_interior, nf):
    """Per-sample tolerance for the inverse-CDF stage, from the oracle's own cdf.

    z = b_lo + (u - cdf_lo)/denom * (b_hi - b_lo): an error eps in the cdf (two fp32 implementations
    differ by a few ulp of 1.0 after a 62-term running sum) moves z by (b_hi-b_lo)*eps/denom, which is
    1e-8 for a bin holding real mass and 3e-3 for a near-empty bin whose denom sits just above the 1e-5
    guard.  Samples whose denom is within 5 % of the guard itself (render.py:52 switches denom -> 1 there)
    or whose u touches a cdf entry can pick the other branch and are masked; the reference against
    itself in fp64 shows the same jumps.  Returns (mask[N,nf], tol[N,nf])."""
    w = torch.as_tensor(weights_interior) + 1e-5
    bins = torch.as_tensor(bins)
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.0, 1.0, steps=nf).expand(cdf.shape[0], nf).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    denom = torch.gather(cdf, -1, hi) - torch.gather(cdf, -1, lo)
    width = torch.gather(bins, -1, hi) - torch.gather(bins, -1, lo)
    eps = 5e-7
    mask = (denom - 1e-5).abs() < 5e-7
    mask |= ((u - torch.gather(cdf, -1, lo)).abs() < eps) | ((u - torch.gather(cdf, -1, hi)).abs() < eps)
    used = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    tol = 2e-6 + width * eps / used
    return mask.numpy(), tol.numpy()


def check_fine_depths(case, zs, zf, ref_s, ref_f, bins, w_interior, nf, stage="sample_fine"):
    """HIP resampled depths zs[N,nf] / merged zf[N,Nc+nf] against the oracle's on the same weights: every sample
    within its conditioning tolerance unless masked; well-conditioned rays within 1e-5 after the merge.
    Returns the fraction of samples that really took the other branch."""
    zs, zf, ref_s, ref_f = _np64(zs), _np64(zf), _np64(ref_s), _np64(ref_f)
    assert (zf[:, 1:] >= zf[:, :-1]).all()
    if nf == 0:
        assert np.array_equal(zf, ref_f)
        record(case=case, stage=stage, qty="z_fine", err_vs_oracle32=0.0, tol=0.0, active="bit-exact", passed=True)
        return 0.0
    mask, tol = pdf_conditioning(bins, w_interior, nf)
    d = np.abs(zs - ref_s)
    bad = (d > tol) & ~mask
    # u = 0 and u = 1 touch the cdf's ends on EVERY ray.  u = 0 is deterministic (cdf[0] is an exact 0); u = 1 against
    # cdf[-1] = 1 +- 1 ulp may bracket either way, which is harmless (t ~ 1 - 1e-7/denom) unless the last bin is
    # near-empty, where the denom < 1e-5 guard turns the two brackets into the two ends of that bin
    w = np.asarray(w_interior, np.float64) + 1e-5
    last_mass = w[:, -1] / w.sum(-1)
    last_width = _np64(bins)[:, -1] - _np64(bins)[:, -2]
    well = (tol <= 1e-5).all(-1) & ~mask[:, 1:-1].any(-1) & (last_width * 5e-7 / last_mass <= 8e-6)
    e_well = float(np.abs(zf[well] - ref_f[well]).max(initial=0.0))
    flips = float((mask & (d > tol)).mean())
    record(case=case, stage=stage, qty="z_samples", err_vs_oracle32=float((d / tol)[~mask].max(initial=0.0)), tol=1.0,
           unit="multiples of the per-sample conditioning tolerance", active="hard", passed=not bad.any(),
           well_conditioned_rays=int(well.sum()), rays=int(well.size), z_fine_err_well=e_well, branch_flip_frac=flips)
    assert not bad.any(), (case, d[bad].max(), tol[bad].min(), int(bad.sum()))
    assert e_well <= 1e-5, (case, e_well)
    return flips


# ----------------------------------------------------------------------------------------------------------------
# render_rays end to end, by composition
# ----------------------------------------------------------------------------------------------------------------
def check_render(case, hip, ref: R.RenderTrace, fields64, rays, near, far, nc, nf, t_rand, coarse_field, fine_field,
                 sharp=False, check_e2e=True) -> dict:
    """`hip`: dict of the HIP path's tensors for this call - the six outputs of the fused render_rays call
    (rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f) and the staged chain's intermediates (weights_c, z_samples,
    z_fine) whose outputs the caller already asserted bit-equal to the fused call.  `ref`: the fp32 oracle trace
    (fixture or oracle run); fields64 = (coarse, fine) oracle fields holding fp64 weights, or None for cases gated
    flat (no floor term: every twin that is not `sharp`).  coarse_field / fine_field: fp32 oracle fields."""
    rays_t, tr = torch.as_tensor(rays), torch.as_tensor(t_rand)
    t64 = i64 = None
    if fields64 is not None:
        with torch.no_grad():
            t64 = R.render_rays_f64(rays_t, near, far, fields64[0], fields64[1], nc, nf, tr)
    g64 = (lambda k: None) if (t64 is None or not sharp) else (lambda k: getattr(t64, k))
    # 1. coarse pass: same inputs as the oracle by construction (bit-exact rays and depths)
    gate(case, "coarse", "rgb", hip["rgb_c"], ref.rgb_c, g64("rgb_c"))
    gate(case, "coarse", "acc", hip["acc_c"], ref.acc_c, g64("acc_c"))
    gate(case, "coarse", "depth", hip["depth_c"], ref.depth_c, g64("depth_c"), tol=DEPTH_TOL)
    # the weights are an intermediate (they only steer the resampling, which step 2 checks on THESE weights): flat
    # gate where it holds, otherwise no further from fp64 than the fp32 oracle
    gate(case, "coarse", "weights", hip["weights_c"], ref.weights_c, None if t64 is None else t64.weights_c,
         factor=FP64_FACTOR_INTERMEDIATE)
    # 2. resampling on the HIP path's own coarse weights against the oracle's sample_pdf on those weights
    n = rays_t.shape[0]
    lin = torch.linspace(near, far, nc)
    mids = (0.5 * (lin[1:] + lin[:-1])).expand(n, nc - 1)
    w_hip = torch.as_tensor(_np64(hip["weights_c"]).astype(np.float32))
    ref_s = R.sample_pdf(mids, w_hip[:, 1:-1], nf)
    ref_zf = torch.sort(torch.cat([ref.z_coarse, ref_s], -1), -1).values
    flips = check_fine_depths(case, hip["z_samples"], hip["z_fine"], ref_s, ref_zf, mids, w_hip[:, 1:-1], nf,
                              stage="resample(HIP weights)")
    # 3. fine pass at the HIP path's own depths: oracle (fp32, and fp64 for sharp heads) evaluated there, every ray
    z_hip = torch.as_tensor(_np64(hip["z_fine"]).astype(np.float32))
    with torch.no_grad():
        at = R.render_rays(rays_t, near, far, coarse_field, fine_field, nc, nf, tr, z_hip)
        if fields64 is not None and sharp:
            i64 = R.render_rays_f64(rays_t, near, far, fields64[0], fields64[1], nc, nf, tr, z_hip)
    h64 = (lambda k: None) if i64 is None else (lambda k: getattr(i64, k))
    gate(case, "fine(HIP depths)", "rgb", hip["rgb_f"], at.rgb_f, h64("rgb_f"))
    gate(case, "fine(HIP depths)", "acc", hip["acc_f"], at.acc_f, h64("acc_f"))
    gate(case, "fine(HIP depths)", "depth", hip["depth_f"], at.depth_f, h64("depth_f"), tol=DEPTH_TOL)
    # 4. end to end against the oracle's own image: distribution, next to the fp32 oracle's distance from fp64
    d = np.abs(_np64(hip["rgb_f"]) - _np64(ref.rgb_f)).max(-1)
    over = d > TOL
    dz = np.abs(_np64(hip["z_fine"]) - _np64(ref.z_fine)).max(-1)
    rec = dict(case=case, stage="end-to-end fine", qty="rgb", err_vs_oracle32=float(d.max()), tol=TOL,
               frac_rays_over=float(over.mean()), psnr_vs_oracle=R.psnr(_np64(hip["rgb_f"]), _np64(ref.rgb_f)),
               branch_flip_frac=flips, active="distribution")
    if t64 is not None:
        d64 = np.abs(_np64(t64.rgb_f) - _np64(ref.rgb_f)).max(-1)
        rec.update(oracle32_vs_fp64=float(d64.max()), frac_rays_over_fp64=float((d64 > TOL).mean()))
    # every ray over the gate is a ray whose fine depths differ from the oracle's (the resampling moved them):
    # where the depths agree the fine pass was just shown to be within the gate
    unexplained = over & (dz == 0.0)
    rec.update(rays_over_with_identical_depths=int(unexplained.sum()))
    ok = True
    if check_e2e:
        ok = (not unexplained.any() or sharp) and rec["frac_rays_over"] <= rec.get("frac_rays_over_fp64", 0.0) + 0.02
    rec["passed"] = bool(ok)
    record(**rec)
    assert ok, rec
    return rec


def hip_stage_chain(ops, pf_c, pf_f, rays, near, far, nc, nf, t_rand, film=None) -> dict:
    """The HIP stages of render_rays chained one C-ABI call at a time (`ops` = mirender.ops, handed in by the caller:
    this package never imports the product).  Same kernels and inputs as the fused mi_render_rays call, so its six
    outputs must equal that call's bit for bit; the intermediates are what `check_render` gates."""
    dev = rays.device
    n = rays.shape[0]
    z_c = ops.sample_coarse(n, near, far, nc, dev, t_rand)
    raw_c = ops.field_eval_rays(pf_c, rays, z_c, film)
    rgb_c, depth_c, acc_c, w_c = ops.composite(raw_c, z_c, rays)
    z_f, z_s = ops.sample_fine(z_c, w_c, near, far, nf, want_samples=True)
    raw_f = ops.field_eval_rays(pf_f, rays, z_f, film)
    rgb_f, depth_f, acc_f, _ = ops.composite(raw_f, z_f, rays, want_weights=False)
    return dict(rgb_c=rgb_c, depth_c=depth_c, acc_c=acc_c, rgb_f=rgb_f, depth_f=depth_f, acc_f=acc_f, z_coarse=z_c,
                raw_c=raw_c, weights_c=w_c, z_samples=z_s, z_fine=z_f, raw_f=raw_f)


def assert_chain_equals_fused(chain: dict, fused) -> None:
    for k, t in zip(("rgb_c", "depth_c", "acc_c", "rgb_f", "depth_f", "acc_f"), fused):
        assert torch.equal(chain[k], t.detach()), f"staged chain and fused render_rays disagree on {k}"
