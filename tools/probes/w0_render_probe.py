"""GPU probe: where does the w_0 = 41.5 render-level gradient mismatch come from?  Stage by stage against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import numpy as np, torch
from mirender import fields, ops, autograd as A
from oracle import fields as ofields, render_ref as R, synth, parity
dev = torch.device("cuda", 0)
n, nc, nf = 128, 8, 16
sd = synth.state_dict("film_siren_nerf", seed=44, sharp="medium")
film0 = synth.film_params(1, seed=6)
rays = torch.from_numpy(R.rays_from_camera(16, 16, 76.0, synth.pose_radians(1.0, 0.15, -0.1))[40:40 + n])
tr = synth.t_rand(n, nc, seed=2)
rng = np.random.Generator(np.random.PCG64(7))
cot = [torch.from_numpy(rng.normal(size=s).astype(np.float32)) for s in ((n, 3), (n,), (n,))]
for w in (30.0, 41.5):
    ofields.W0 = w
    m = fields.FilmSirenNeRF(w_0=w).to(dev); m.load_state_dict(sd)
    pf = fields.as_packed_field(m)
    ch = parity.hip_stage_chain(ops, pf, pf, rays.to(dev), 0.5, 1.5, nc, nf, tr.to(dev), film0.to(dev))
    z = ch["z_fine"]
    raw = ch["raw_f"]
    # oracle raw at the same z (fp32 / fp64)
    res = {}
    for dt in (torch.float32, torch.float64):
        fo = ofields.make_field("film_siren_nerf", {k: v.to(dt) for k, v in sd.items()}, film0[0].to(dt))
        r = rays.to(dt); view = r[:, 1] / torch.norm(r[:, 1], dim=-1, keepdim=True)
        ro = R.query_field(R.points_on_rays(r[:, 0], r[:, 1], z.cpu().to(dt)), view, fo).detach().requires_grad_(True)
        out = R.composite(ro, z.cpu().to(dt), r[:, 1])
        sum((o * c.to(dt)).sum() for o, c in zip(out[:3], cot)).backward()
        res[dt] = (ro.detach().double(), ro.grad.double())
    g_hip = A._composite_bwd(raw, z, rays.to(dev), *[c.to(dev) for c in cot]).double().cpu()
    # composite bwd on the ORACLE's fp32 raw through the HIP kernel: isolates the kernel from the raw values
    g_hip_on_oracle_raw = A._composite_bwd(res[torch.float32][0].float().to(dev), z, rays.to(dev), *[c.to(dev) for c in cot]).double().cpu()
    r64, g64 = res[torch.float64]; r32, g32 = res[torch.float32]
    print(f"w_0={w}: raw_f max|hip-64| {float((raw.double().cpu()-r64).abs().max()):.2e}  max|cpu32-64| {float((r32-r64).abs().max()):.2e}; sigma max {float(r64[...,3].max()):.1f}")
    for name, g in (("hip", g_hip), ("hip kernel on oracle32 raw", g_hip_on_oracle_raw), ("cpu32", g32)):
        e = (g - g64)
        print(f"    g_raw {name:28s} rel l2 {float(e.norm()/g64.norm()):.2e}  sigma-channel rel {float(e[...,3].norm()/g64[...,3].norm()):.2e} rgb-channels rel {float(e[...,:3].norm()/g64[...,:3].norm()):.2e}")
    k = (g_hip - g64)[..., 3].abs().reshape(-1).argmax()
    ray, smp = int(k) // (nc + nf), int(k) % (nc + nf)
    print("    worst sigma-grad element: ray", ray, "sample", smp, "hip", float(g_hip[ray, smp, 3]), "f64", float(g64[ray, smp, 3]), "cpu32", float(g32[ray, smp, 3]),
          "sigma hip/64/32", float(raw[ray, smp, 3]), float(r64[ray, smp, 3]), float(r32[ray, smp, 3]), "dz", float(z[ray, min(smp+1, nc+nf-1)] - z[ray, smp]))

print("---- render-level gradients, w_0 = 41.5")
from mirender import render_core
w = 41.5
ofields.W0 = w
cot6 = [torch.from_numpy(rng.normal(size=s).astype(np.float32)) for s in ((n, 3), (n,), (n,), (n, 3), (n,), (n,))]
def oracle(z_f, which):
    out = {}
    for dt in (torch.float32, torch.float64):
        sdr = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
        fo = ofields.make_field("film_siren_nerf", sdr, film0[0].to(dt))
        t = R.render_rays(rays.to(dt), 0.5, 1.5, fo, fo, nc, nf, tr.to(dt), z_f.to(dt))
        loss = sum((t.outputs()[i] * cot6[i].to(dt)).sum() for i in which)
        loss.backward()
        out[dt] = {k: v.grad.double() for k, v in sdr.items()}
    return out
for label, which, shared in (("all six, shared", range(6), True), ("rgb_f only, shared", [3], True), ("rgb_c only, shared", [0], True),
                             ("acc_f only, shared", [5], True), ("all six, two views", range(6), False)):
    m = fields.FilmSirenNeRF(w_0=w).to(dev); m.load_state_dict(sd)
    pf = fields.as_packed_field(m)
    pf2 = pf if shared else fields.PackedField(pf.kind, pf.params, w)
    film = film0.to(dev)
    out = A.render_rays_train(pf, pf2, rays.to(dev), 0.5, 1.5, nc, nf, film, tr.to(dev), 0)
    sum((out[i] * cot6[i].to(dev)).sum() for i in which).backward()
    ch = parity.hip_stage_chain(ops, pf, pf, rays.to(dev), 0.5, 1.5, nc, nf, tr.to(dev), film)
    ref = oracle(ch["z_fine"].cpu(), which)
    errs = {k: (float((p.grad.double().cpu() - ref[torch.float64][k]).norm() / ref[torch.float64][k].norm()),
                float((ref[torch.float32][k] - ref[torch.float64][k]).norm() / ref[torch.float64][k].norm())) for k, p in m.named_parameters()}
    wk = max(errs, key=lambda k: errs[k][0])
    print(f"  {label:22s} worst {wk:28s} hip {errs[wk][0]:.2e} cpu32 {errs[wk][1]:.2e} | sigma head w {errs['output_layer_sigma.0.weight'][0]:.2e} rgb head w {errs['output_layer_rgb.0.weight'][0]:.2e}")

print("---- composite_bwd per entry, sigma > 0 entries only, rgb cotangent only")
for w in (30.0, 41.5):
    ofields.W0 = w
    m = fields.FilmSirenNeRF(w_0=w).to(dev); m.load_state_dict(sd)
    pf = fields.as_packed_field(m)
    ch = parity.hip_stage_chain(ops, pf, pf, rays.to(dev), 0.5, 1.5, nc, nf, tr.to(dev), film0.to(dev))
    z, raw = ch["z_fine"], ch["raw_f"]
    res = {}
    for dt in (torch.float32, torch.float64):
        ro = raw.cpu().to(dt).requires_grad_(True)            # the SAME raw values (HIP's) through the oracle's compositing
        out = R.composite(ro, z.cpu().to(dt), rays[:, 1].to(dt))
        (out[0] * cot[0].to(dt)).sum().backward()
        res[dt] = ro.grad.double()
    g = A._composite_bwd(raw, z, rays.to(dev), cot[0].to(dev), None, None).double().cpu()
    mask = raw.cpu()[..., 3] > 0
    g64, g32 = res[torch.float64], res[torch.float32]
    for name, gg in (("hip", g), ("cpu32", g32)):
        e = (gg - g64)[..., 3][mask]
        ref = g64[..., 3][mask]
        print(f"  w_0={w} {name:6s} sigma-grad on sigma>0 entries: rel l2 {float(e.norm()/ref.norm()):.2e} max|err| {float(e.abs().max()):.2e} max|ref| {float(ref.abs().max()):.2e} sum err {float(e.sum()):.2e} sum ref {float(ref.sum()):.2e}")
    k = int(((g - g64)[..., 3].abs() * mask).reshape(-1).argmax()); ray, smp = k // (nc + nf), k % (nc + nf)
    print("     worst entry: ray", ray, "sample", smp, "hip", float(g[ray, smp, 3]), "f64", float(g64[ray, smp, 3]), "cpu32", float(g32[ray, smp, 3]))
    print("     that ray's sigma:", [round(float(v), 3) for v in raw[ray, :, 3].cpu()])

print("---- sigma-head ReLU switches: HIP vs fp64 oracle vs fp32 oracle at the same depths")
for w in (30.0, 41.5):
    ofields.W0 = w
    m = fields.FilmSirenNeRF(w_0=w).to(dev); m.load_state_dict(sd)
    pf = fields.as_packed_field(m)
    ch = parity.hip_stage_chain(ops, pf, pf, rays.to(dev), 0.5, 1.5, nc, nf, tr.to(dev), film0.to(dev))
    for key, zz in (("coarse", ch["z_coarse"]), ("fine", ch["z_fine"])):
        sig = {}
        for dt in (torch.float32, torch.float64):
            fo = ofields.make_field("film_siren_nerf", {k: v.to(dt) for k, v in sd.items()}, film0[0].to(dt))
            r = rays.to(dt); view = r[:, 1] / torch.norm(r[:, 1], dim=-1, keepdim=True)
            sig[dt] = R.query_field(R.points_on_rays(r[:, 0], r[:, 1], zz.cpu().to(dt)), view, fo)[..., 3]
        hip = ch["raw_c" if key == "coarse" else "raw_f"][..., 3].cpu()
        f_h = int(((hip > 0) != (sig[torch.float64] > 0)).sum()); f_c = int(((sig[torch.float32] > 0) != (sig[torch.float64] > 0)).sum())
        where = ((hip > 0) != (sig[torch.float64] > 0)).nonzero().tolist()
        print(f"  w_0={w} {key}: switches that differ from fp64: hip {f_h}, cpu32 {f_c}", where, [float(sig[torch.float64][a, b]) for a, b in where], [float(hip[a, b]) for a, b in where])
