"""GPU: the generic path (any callable, a same-layout look-alike with another w_0, a mixed pair) and the public stage
functions are differentiable like the reference's torch ops (nerf/render.py:59-103, 106-147; pi_GAN/synthesis.py:83-107
optimises FiLM leaves through them).

The reference's `render_rays` / `raw_to_outputs` / `run_network` are plain torch and carry gradients for ANY model.  Here
the callable's own autograd graph reaches `raw`, compositing continues it on the HIP kernels (mi_composite /
mi_composite_bwd behind a torch.autograd.Function) and the fused side of a mixed pair goes through its own saving
forward / backward chain / dW GEMMs, one pass at a time.  Every gradient tensor is gated against the oracle's autograd
in fp64 (oracle/parity.py:gate_grad) at the HIP path's OWN fine depths - hierarchical resampling is ill-conditioned
(DESIGN.md section 2), so the depths the HIP call really used are recorded and injected into the oracle - and the records
land in the session's parity file."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fields as ofields, fit_ref, parity, render_ref as R, synth  # noqa: E402


def dev():
    return torch.device("cuda", 0)


class SmallField(torch.nn.Module):
    """A user's own field: network([M,6]) -> [M,4] (rgb in [0,1], sigma >= 0), smooth hidden activations so that its
    gradients have no 0/1 switches of their own.  Plain torch: runs on the device for the HIP path, on the CPU (fp32 and
    fp64 copies) for the oracle."""

    def __init__(self, seed, width=64):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.l0, self.l1, self.l2 = torch.nn.Linear(6, width), torch.nn.Linear(width, width), torch.nn.Linear(width, 4)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (1.0 / np.sqrt(p.shape[-1])))
            self.l0.weight.mul_(3.0)
            self.l2.bias[3] = 0.5

    def forward(self, x):
        h = torch.tanh(self.l1(torch.sin(self.l0(x))))
        o = self.l2(h)
        return torch.cat([torch.sigmoid(o[:, :3]), torch.relu(o[:, 3:] * 4)], -1)


class _FilmSirenLayer(torch.nn.Module):                      # the reference layer's arithmetic (pi_GAN/modules.py:22-25)
    def __init__(self, i, o, w_0, named):
        super().__init__()
        # `named`: carries the reference layer's `w_0` attribute (pi_GAN/modules.py:16) - what the fused kernels read the
        # frequency from.  Otherwise the same number under another name: the layer does not say what it computes, so the
        # module is NOT claimed and renders through the generic path (its own forward).
        setattr(self, "w_0" if named else "omega", w_0)
        self.weight = torch.nn.Parameter(torch.zeros(o, i))
        self.bias = torch.nn.Parameter(torch.zeros(o))

    def forward(self, x, gamma, beta):
        w = self.w_0 if hasattr(self, "w_0") else self.omega
        return torch.sin(w * (gamma * torch.nn.functional.linear(x, self.weight, self.bias) + beta))


class FilmLookAlike(torch.nn.Module):
    """pi_GAN/modules.py:70-118 with w_0 as the constructor argument it is there (:73): the FiLM layout; claimed by the
    fused kernels - whatever w_0 - when its layers name their w_0 like the reference's, generic path otherwise."""

    def __init__(self, w_0, named=True):
        super().__init__()
        self.film_params = None
        self.input_layer = _FilmSirenLayer(3, 256, w_0, named)
        self.hidden_layers = torch.nn.ModuleList([_FilmSirenLayer(256, 256, w_0, named) for _ in range(7)])
        self.output_layer_sigma = torch.nn.Sequential(torch.nn.Linear(256, 1), torch.nn.ReLU())
        self.hidden_layer_rgb = _FilmSirenLayer(259, 256, w_0, named)
        self.output_layer_rgb = torch.nn.Sequential(torch.nn.Linear(256, 3), torch.nn.Sigmoid())

    def forward(self, x):
        fp = self.film_params
        pos, d = x[:, :3], x[:, 3:]
        h = self.input_layer(pos, *fp[0])
        for i, lay in enumerate(self.hidden_layers):
            h = lay(h, *fp[i + 1])
        sigma = self.output_layer_sigma(h)
        h = self.hidden_layer_rgb(torch.cat([h, d], -1), *fp[8])
        return torch.cat([self.output_layer_rgb(h), sigma], -1)


def _cotangents(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [torch.from_numpy(rng.normal(size=s).astype(np.float32)) for s in ((n, 3), (n,), (n,), (n, 3), (n,), (n,))]


def _hip_render_with_depths(coarse, fine, rays, near, far, nc, nf, tr, monkeypatch, film=None):
    """render_rays on the HIP path with grad; also returns the fine depths that very call used (ops.sample_fine is wrapped
    for the duration of the call - no second evaluation that could round differently)."""
    from mirender import ops, render_core
    seen = {}
    real = ops.sample_fine

    def spy(*a, **k):
        out = real(*a, **k)
        seen["z_fine"] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return out
    monkeypatch.setattr(ops, "sample_fine", spy)
    out = render_core.render_rays(rays.to(dev()), near, far, coarse, fine, nc, nf, t_rand=tr.to(dev()), film=film)
    monkeypatch.setattr(ops, "sample_fine", real)
    assert "z_fine" in seen, "the call did not take the generic path (ops.sample_fine was never called)"
    return out, seen["z_fine"].cpu()


def _oracle_grads(make, rays, near, far, nc, nf, tr, z_fine, cot):
    """{dtype: {name: grad}} from the oracle's autograd at the injected fine depths.  `make(dtype)` -> (coarse callable,
    fine callable, {name: leaf tensor})."""
    out = {}
    for dt in (torch.float32, torch.float64):
        fc, ff, leaves = make(dt)
        t = R.render_rays(rays.to(dt), near, far, fc, ff, nc, nf, tr.to(dt), z_fine.to(dt))
        loss = sum((o * c.to(dt)).sum() for o, c in zip(t.outputs(), cot))
        grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        out[dt] = {k: (torch.zeros_like(v) if g is None else g).double() for (k, v), g in zip(leaves.items(), grads)}
    return out


def _gate_all(case, got: dict, refs, smooth_names=None):
    """One gate_grad record per tensor; ReLU-network tensors (those not in smooth_names) at the ReLU gate."""
    recs = []
    for name, g in got.items():
        smooth = smooth_names is None or name in smooth_names
        recs.append(parity.gate_grad(case, name, g.detach().cpu(), refs[torch.float32][name], refs[torch.float64][name],
                                     tol=parity.GRAD_TOL_SMOOTH if smooth else parity.GRAD_TOL_RELU,
                                     elem_tol=parity.GRAD_ELEM_TOL_SMOOTH if smooth else None, check=False))
    bad = [r for r in recs if not r["passed"]]
    assert not bad, bad[0]
    return recs


def _rays(n, w=24, h=24, pose=None, focal=33.0, first=60):
    pose = synth.pose_degrees(4.0, 15.0, -30.0) if pose is None else pose
    return torch.from_numpy(R.rays_from_camera(w, h, focal, pose)[first:first + n])


def test_callable_pair_every_parameter_gradient(monkeypatch):
    """(i) two SmallField callables through render_rays: all six outputs carry a cotangent; every parameter gradient of
    both models against the oracle's autograd (the same modules on the CPU, fp32 and fp64) at the HIP call's own depths."""
    n, nc, nf = 200, 12, 20
    rays, tr, cot = _rays(n), synth.t_rand(n, nc, seed=3), _cotangents(n, 5)
    cm, fm = SmallField(1).to(dev()), SmallField(2).to(dev())
    out, z_f = _hip_render_with_depths(cm, fm, rays, 2.0, 6.0, nc, nf, tr, monkeypatch)
    assert all(o.requires_grad for o in out)
    sum((o * c.to(dev())).sum() for o, c in zip(out, cot)).backward()
    got = {f"coarse.{k}": p.grad for k, p in cm.named_parameters()}
    got.update({f"fine.{k}": p.grad for k, p in fm.named_parameters()})
    assert all(g is not None and torch.isfinite(g).all() for g in got.values())

    def make(dt):
        c, f = SmallField(1).to(dt), SmallField(2).to(dt)
        leaves = {f"coarse.{k}": p for k, p in c.named_parameters()}
        leaves.update({f"fine.{k}": p for k, p in f.named_parameters()})
        return c, f, leaves
    refs = _oracle_grads(make, rays, 2.0, 6.0, nc, nf, tr, z_f, cot)
    # forward values at those depths too: the flat 1e-4 gate (depth 5e-4)
    with torch.no_grad():
        c32, f32, _ = make(torch.float32)
        t = R.render_rays(rays, 2.0, 6.0, c32, f32, nc, nf, tr, z_f)
    case = f"generic path: SmallField callable pair {n} rays {nc}+{nf}"
    for name, g, r, tol in (("rgb_c", out[0], t.rgb_c, parity.TOL), ("depth_c", out[1], t.depth_c, parity.DEPTH_TOL),
                            ("rgb_f", out[3], t.rgb_f, parity.TOL), ("acc_f", out[5], t.acc_f, parity.TOL)):
        parity.gate(case, "forward (HIP depths)", name, g.detach().cpu(), r, tol=tol)
    _gate_all(case, got, refs)


@pytest.mark.parametrize("w_0,named", [(25.0, False), (25.0, True), (30.0, True), (36.0, True)])
def test_film_look_alike_weights_and_film_leaves(monkeypatch, w_0, named):
    """(ii) the FiLM layout with w_0 = 25 through the generic path (layers that do not name their w_0: the module's own
    forward) and w_0 in {25, 30, 41.5} claimed by the fused kernels, which take FilmSiren's w_0 (pi_GAN/modules.py:11,73) at
    run time: a loss on the fine image, gradients of every weight AND of the FiLM leaves - the shape of
    pi_GAN/synthesis.py:83-107, which optimises gamma / beta directly - against the oracle evaluated with that w_0.
    (A FiLM field is smooth except for its sigma head, a ReLU.  This fixture's weights are initialised for w_0 = 30, so a
    larger w_0 amplifies rounding through the eight sin layers - at 41.5 every fp32 pipeline's sigma is 1.2e-4 from fp64 -
    and with 41.5 one sample of one ray has sigma_pre within that distance of 0: the HIP path and the fp64 oracle then
    disagree about ONE 0/1 switch, which alone moves the sigma head's bias gradient by 1 % (tools/probes/w0_render_probe.py;
    points-mode gradients at 41.5 and 50 agree to 3e-5 / 8e-5, tools/probes/w0_probe.py).  33, 36 and 38 have no such
    sample; 36 is kept.)"""
    from mirender import fields
    n, nc, nf = 128, 8, 16
    sd = synth.state_dict("film_siren_nerf", seed=44, sharp="medium")
    film0 = synth.film_params(1, seed=6)[0]                               # [9,512]
    rays = torch.from_numpy(R.rays_from_camera(16, 16, 76.0, synth.pose_radians(1.0, 0.15, -0.1))[40:40 + n])
    tr, cot = synth.t_rand(n, nc, seed=2), _cotangents(n, 7)
    cot[1], cot[2], cot[4] = torch.zeros(n), torch.zeros(n), torch.zeros(n)   # pi_GAN consumes rgb (and acc) only
    m = FilmLookAlike(w_0, named).to(dev())
    m.load_state_dict(sd)
    film = film0.to(dev()).requires_grad_(True)
    m.film_params = [torch.chunk(film[i], 2) for i in range(9)]
    pf = fields.as_packed_field(m)
    assert (pf is None) == (not named) and (pf is None or pf.w_0 == w_0)
    if pf is None:
        out, z_f = _hip_render_with_depths(m, m, rays, 0.5, 1.5, nc, nf, tr, monkeypatch)
    else:                                                                 # fused pair: depths from the stage chain
        from mirender import ops, render_core
        out = render_core.render_rays(rays.to(dev()), 0.5, 1.5, m, m, nc, nf, t_rand=tr.to(dev()))
        with torch.no_grad():
            ch = parity.hip_stage_chain(ops, pf, pf, rays.to(dev()), 0.5, 1.5, nc, nf, tr.to(dev()), film.detach().reshape(1, 9, 512))
        z_f = ch["z_fine"].cpu()
    sum((o * c.to(dev())).sum() for o, c in zip(out, cot)).backward()
    got = {k: p.grad for k, p in m.named_parameters()}
    got["__film__"] = film.grad
    assert film.grad is not None and float(film.grad.abs().max()) > 0
    monkeypatch.setattr(ofields, "W0", w_0)

    def make(dt):
        sd_req = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
        f_req = film0.clone().to(dt).requires_grad_(True)
        fo = ofields.make_field("film_siren_nerf", sd_req, f_req)
        return fo, fo, dict(sd_req, __film__=f_req)
    refs = _oracle_grads(make, rays, 0.5, 1.5, nc, nf, tr, z_f, cot)
    _gate_all(f"FiLM look-alike w_0={w_0:g} {n} rays {nc}+{nf} ({'generic path' if pf is None else 'fused kernels'}): "
              "weights and FiLM leaves", got, refs)


@pytest.mark.parametrize("fused_side,kind", [("coarse", "nerf"), ("fine", "siren_nerf"), ("coarse", "film_siren_nerf")])
def test_mixed_pair(monkeypatch, fused_side, kind):
    """(iii) one fused kind and one callable in the same render_rays call: the fused side's parameters (and FiLM leaves)
    get their gradients from its own backward kernels, the callable's from torch autograd, both against the oracle."""
    from mirender import fields
    n, nc, nf = 192, 12, 20
    is_film = kind.startswith("film")
    near, far = (0.5, 1.5) if is_film else (2.0, 6.0)
    rays = _rays(n) if not is_film else torch.from_numpy(R.rays_from_camera(16, 16, 76.0, synth.pose_radians(1.0, 0.15, -0.1))[20:20 + n])
    tr, cot = synth.t_rand(n, nc, seed=4), _cotangents(n, 9)
    sd = synth.state_dict(kind, seed=21, sharp="medium", bias_jitter=0.05)
    fused = fields.field_from_state_dict(sd, dev())
    film0 = synth.film_params(1, seed=12) if is_film else None            # [1,9,512]
    film = film0.to(dev()).requires_grad_(True) if is_film else None
    call = SmallField(3).to(dev())
    cm, fm = (fused, call) if fused_side == "coarse" else (call, fused)
    out, z_f = _hip_render_with_depths(cm, fm, rays, near, far, nc, nf, tr, monkeypatch, film=film)
    sum((o * c.to(dev())).sum() for o, c in zip(out, cot)).backward()
    got = {f"fused.{k}": p.grad for k, p in fused.named_parameters()}
    got.update({f"callable.{k}": p.grad for k, p in call.named_parameters()})
    if is_film:
        got["__film__"] = film.grad.reshape(9, 512)
    assert all(g is not None for g in got.values())

    def make(dt):
        sd_req = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
        f_req = film0[0].clone().to(dt).requires_grad_(True) if is_film else None
        fo = ofields.make_field(kind, sd_req, f_req)
        c = SmallField(3).to(dt)
        leaves = {f"fused.{k}": v for k, v in sd_req.items()}
        leaves.update({f"callable.{k}": p for k, p in c.named_parameters()})
        if is_film:
            leaves["__film__"] = f_req
        return ((fo, c) if fused_side == "coarse" else (c, fo)) + (leaves,)
    refs = _oracle_grads(make, rays, near, far, nc, nf, tr, z_f, cot)
    smooth = None if kind != "nerf" else {k for k in got if k.startswith("callable.")}
    _gate_all(f"mixed pair: fused {kind} as the {fused_side} model + SmallField callable, {n} rays {nc}+{nf}", got, refs, smooth)


def test_eight_sgd_steps_through_the_generic_path_follow_the_oracle_loop(golden):
    """(iv) the loop of nerf/train_nerf.py:124-176 on a callable pair - teacher-scene pictures, all 3 456 rays per step,
    plain SGD (a step proportional to the gradient shows a gradient of the wrong size in the next loss) - on the HIP path
    and on the CPU through the oracle, same modules, same jitter: every step's loss within 1e-4 relative.  The regime's own
    noise (the CPU loop rerun from weights perturbed by 1e-6 relative: 3e-7 per loss; fp32 vs fp64: 1.5e-7) is far below
    the gate; the loss halves in the eight steps."""
    from mirender import render_core
    scene = fit_ref.Scene(student="tiny_nerf", images=golden("fit_r03_scene")["images"])
    steps, lr = 8, 0.5

    def loop(render, to_dev, cm, fm):
        opt = torch.optim.SGD(list(cm.parameters()) + list(fm.parameters()), lr=lr)
        losses = []
        for step in range(steps):
            rays, rgb, tr = scene.batch(step, 0)
            rgb = to_dev(rgb)
            out = render(to_dev(rays), cm, fm, to_dev(tr))
            loss = torch.mean((out[3] - rgb) ** 2) + torch.mean((out[0] - rgb) ** 2)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return np.array(losses)
    cpu = loop(lambda r, c, f, t: R.render_rays(r, fit_ref.NEAR, fit_ref.FAR, c, f, scene.nc, scene.nf, t).outputs(),
               lambda t: t, SmallField(1), SmallField(2))
    hip = loop(lambda r, c, f, t: render_core.render_rays(r, fit_ref.NEAR, fit_ref.FAR, c, f, scene.nc, scene.nf, t_rand=t),
               lambda t: t.to(dev()), SmallField(1).to(dev()), SmallField(2).to(dev()))
    rel = np.abs(hip - cpu) / cpu
    ok = bool(cpu[-1] < 0.55 * cpu[0] and rel.max() <= 1e-4)
    parity.record(case=f"generic path: SmallField callable pair fitted to the teacher scene, {steps} sgd steps (lr {lr}) of all 3456 rays",
                  stage="training trajectory", qty="loss per step", err_vs_oracle32=float(rel.max()), tol=1e-4,
                  unit="max relative loss difference over the steps", first_loss=float(cpu[0]), final_loss_hip=float(hip[-1]),
                  final_loss_cpu=float(cpu[-1]), cpu_side="oracle loop (render_ref.render_rays + torch.optim.SGD) run live",
                  active="hard", passed=ok)
    assert cpu[-1] < 0.55 * cpu[0]
    assert rel.max() <= 1e-4, (int(rel.argmax()), float(rel.max()))


def test_public_stage_functions_are_differentiable():
    """raw_to_outputs (render.py:78-103) returns four tensors, all differentiable with respect to raw - the weights too -
    and run_network / a fused module called on its own (render.py:59-75) carry gradients to the parameters; depths or rays
    that require grad raise instead of silently dropping the gradient."""
    from mirender import _lib, fields, render_core
    rng = np.random.Generator(np.random.PCG64(11))
    n, S = 130, 40
    raw = rng.uniform(0, 1, size=(n, S, 4)).astype(np.float32)
    raw[..., 3] = rng.exponential(2.0, size=(n, S)).astype(np.float32) * (rng.random((n, S)) < 0.6)
    z = np.sort(rng.uniform(2, 6, size=(n, S)).astype(np.float32), -1)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    cot = [torch.from_numpy(rng.normal(size=s).astype(np.float32)) for s in ((n, 3), (n,), (n,), (n, S))]
    ref = {}
    for dt in (torch.float32, torch.float64):
        rt = torch.from_numpy(raw).to(dt).requires_grad_(True)
        outs = R.composite(rt, torch.from_numpy(z).to(dt), torch.from_numpy(rd).to(dt))
        sum((o * c.to(dt)).sum() for o, c in zip(outs, cot)).backward()
        ref[dt] = rt.grad.double()
    rt = torch.from_numpy(raw).to(dev()).requires_grad_(True)
    outs = render_core.raw_to_outputs(rt, torch.from_numpy(z).to(dev()), torch.from_numpy(rd).to(dev()))
    assert len(outs) == 4 and all(o.requires_grad for o in outs)
    sum((o * c.to(dev())).sum() for o, c in zip(outs, cot)).backward()
    parity.gate_grad(f"raw_to_outputs {n} rays S={S}: cotangents on rgb, depth, acc AND weights", "raw", rt.grad.cpu(),
                     ref[torch.float32], ref[torch.float64], tol=2e-5, elem_tol=2e-4)
    with pytest.raises(_lib.MiRenderError, match="requires grad"):
        render_core.raw_to_outputs(rt, torch.from_numpy(z).to(dev()).requires_grad_(True), torch.from_numpy(rd).to(dev()))

    # network(x) on its own / run_network: parameter gradients of a fused module
    for kind in ("nerf", "siren_nerf", "film_siren_nerf"):
        sd = synth.state_dict(kind, seed=31, sharp="medium", bias_jitter=0.05)
        m = fields.field_from_state_dict(sd, dev())
        film0 = synth.film_params(1, seed=13)[0] if kind.startswith("film") else None
        M = 300
        x = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, size=(M, 3)), rng.normal(size=(M, 3))], -1).astype(np.float32))
        x[:, 3:] /= x[:, 3:].norm(dim=-1, keepdim=True)
        c4 = torch.from_numpy(rng.normal(size=(M, 4)).astype(np.float32))
        film = None
        if film0 is not None:
            film = film0.to(dev()).requires_grad_(True)
            m.set_film_params(film)
        y = m(x.to(dev()))
        assert y.requires_grad and tuple(y.shape) == (M, 4)
        (y * c4.to(dev())).sum().backward()
        refs = {}
        for dt in (torch.float32, torch.float64):
            sd_req = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
            f_req = film0.clone().to(dt).requires_grad_(True) if film0 is not None else None
            yo = ofields.make_field(kind, sd_req, f_req)(x.to(dt))
            (yo * c4.to(dt)).sum().backward()
            refs[dt] = {k: v.grad.double() for k, v in sd_req.items()}
            if f_req is not None:
                refs[dt]["__film__"] = f_req.grad.double()
        got = {k: p.grad for k, p in m.named_parameters()}
        if film is not None:
            got["__film__"] = film.grad
        _gate_all(f"network(x) on its own: fused {kind} module, {M} free-standing points", got, refs,
                  None if kind != "nerf" else set())
        # the same through run_network (render.py:59-75) equals the module call
        if film0 is None:
            for p in m.parameters():
                p.grad = None
            pts, view = x[:, :3].reshape(M // 4, 4, 3), x[::4, 3:].contiguous()
            r = render_core.run_network(pts.to(dev()), view.to(dev()), m)
            assert r.requires_grad and tuple(r.shape) == (M // 4, 4, 4)
            r.sum().backward()
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        with pytest.raises(_lib.MiRenderError, match="requires grad"):
            m(x.to(dev()).requires_grad_(True))


def test_generic_path_without_grad_is_unchanged_and_detached():
    """Under no_grad (render_image, inference) the generic path returns plain tensors and the same values as with grad."""
    from mirender import render_core
    n, nc, nf = 64, 8, 8
    rays, tr = _rays(n).to(dev()), synth.t_rand(n, nc, seed=3).to(dev())
    cm, fm = SmallField(1).to(dev()), SmallField(2).to(dev())
    a = render_core.render_rays(rays, 2.0, 6.0, cm, fm, nc, nf, t_rand=tr)
    with torch.no_grad():
        b = render_core.render_rays(rays, 2.0, 6.0, cm, fm, nc, nf, t_rand=tr)
    assert all(x.requires_grad for x in a) and not any(x.requires_grad for x in b)
    for x, y in zip(a, b):
        assert torch.equal(x.detach(), y)
    frozen_c, frozen_f = copy.deepcopy(cm).requires_grad_(False), copy.deepcopy(fm).requires_grad_(False)
    c = render_core.render_rays(rays, 2.0, 6.0, frozen_c, frozen_f, nc, nf, t_rand=tr)
    assert not any(x.requires_grad for x in c)


@pytest.mark.parametrize("kind", ["nerf", "film_siren_nerf"])
def test_points_mode_ranges_kept_or_recomputed_give_the_same_gradients(kind, monkeypatch):
    """A fused module called on its own with more points than one backward range holds: the split into ranges (whole FiLM
    groups per range), kept layer inputs and recomputed ones must all give the gradients of the single-range call."""
    from mirender import autograd, fields
    rng = np.random.Generator(np.random.PCG64(5))
    M = 4 * 160
    x = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, size=(M, 3)), rng.normal(size=(M, 3))], -1).astype(np.float32)).to(dev())
    c4 = torch.from_numpy(rng.normal(size=(M, 4)).astype(np.float32)).to(dev())
    m = fields.field_from_state_dict(synth.state_dict(kind, seed=35, sharp="medium", bias_jitter=0.05), dev())
    pf = fields.as_packed_field(m)
    film = synth.film_params(4, seed=14).to(dev()).requires_grad_(True) if kind.startswith("film") else None
    acts_bytes = 4 * autograd._lib.load().mi_field_train_acts_floats(pf.kind)
    results = []
    for per_range, keep in ((1 << 30, 1 << 40), (160, 1 << 40), (160, 0), (160, acts_bytes * 160 * 2)):
        monkeypatch.setattr(autograd, "_max_points_per_chunk", lambda pf_, n=per_range: n)
        monkeypatch.setattr(autograd, "SAVE_FINE_BYTES", keep)
        for p in m.parameters():
            p.grad = None
        if film is not None:
            film.grad = None
        y = autograd.field_eval_points(pf, x, film)
        assert y.requires_grad and tuple(y.shape) == (M, 4)
        (y * c4).sum().backward()
        results.append([y.detach().clone()] + [p.grad.clone() for p in m.parameters()] + ([] if film is None else [film.grad.clone()]))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0])                           # the forward values do not depend on the split
        for a, b in zip(results[0][1:], other[1:]):
            assert float((a - b).abs().max()) <= 2e-5 * max(1e-3, float(a.abs().max()))
    for a, b in zip(results[1][1:], results[2][1:]):                          # same ranges, kept vs recomputed: bit for bit
        assert torch.equal(a, b)
    for a, b in zip(results[1][1:], results[3][1:]):
        assert torch.equal(a, b)
