"""GPU: checkpoint interop - the reference's .tar dicts (nerf/train_nerf.py:181-189,
pi_GAN/train.py:162-172) written from our modules, read back with torch.load(weights_only=True), rebuilt into fused
modules and rendered bit-identically; and the other direction: a state dict in the reference's layout coming from
plain nn.Linear modules (what the reference's classes are made of)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import render_ref as R, synth  # noqa: E402


def dev():
    return torch.device("cuda", 0)


def _rays(n=200):
    return torch.from_numpy(R.rays_from_camera(24, 24, 33.3, synth.pose_degrees(4.0, 25.0, -30.0))[100:100 + n]).to(dev())


def test_nerf_checkpoint_round_trip(tmp_path):
    from mirender import checkpoint, fields, render_core
    cm, fm = fields.NeRF().to(dev()), fields.NeRF().to(dev())
    cm.load_state_dict(synth.state_dict("nerf", 70, "medium", 0.05))
    fm.load_state_dict(synth.state_dict("nerf", 71, "medium", 0.05))
    opt = torch.optim.Adam(list(cm.parameters()) + list(fm.parameters()), lr=5e-4, betas=(0.9, 0.999))
    rays, tr = _rays(), synth.t_rand(200, 16, seed=3).to(dev())
    out = render_core.render_rays(rays, 2.0, 6.0, cm, fm, 16, 24, t_rand=tr)
    (out[3].square().mean() + out[0].square().mean()).backward()
    opt.step()                                            # the optimiser state is part of the checkpoint
    path = checkpoint.checkpoint_path(str(tmp_path), 1234)
    assert os.path.basename(path) == "001234.tar"
    checkpoint.save_nerf(path, 1234, cm, fm, opt)
    open(os.path.join(str(tmp_path), "config.json"), "w").write("{}")       # not a checkpoint: no 'tar' in its name
    checkpoint.save_nerf(checkpoint.checkpoint_path(str(tmp_path), 200), 200, cm, fm, opt)
    assert checkpoint.latest(str(tmp_path)) == path       # lexicographically last, train_nerf.py:101-104
    ck = checkpoint.load(path)                            # weights_only=True
    assert set(ck) == {"global_step", "coarse_model", "fine_model", "optimizer"} and ck["global_step"] == 1234
    assert list(ck["coarse_model"]) == list(cm.state_dict())
    cm2, fm2 = checkpoint.nerf_models(ck, dev())
    opt2 = torch.optim.Adam(list(cm2.parameters()) + list(fm2.parameters()), lr=5e-4, betas=(0.9, 0.999))
    opt2.load_state_dict(ck["optimizer"])                 # train_nerf.py:110
    with torch.no_grad():
        a = render_core.render_rays(rays, 2.0, 6.0, cm, fm, 16, 24, t_rand=tr)
        b = render_core.render_rays(rays, 2.0, 6.0, cm2, fm2, 16, 24, t_rand=tr)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # resumed training continues identically: one more step on both
    for (c_, f_, o_) in ((cm, fm, opt), (cm2, fm2, opt2)):
        o_.zero_grad()
        out = render_core.render_rays(rays, 2.0, 6.0, c_, f_, 16, 24, t_rand=tr)
        (out[3].square().mean() + out[0].square().mean()).backward()
        o_.step()
    for p, q in zip(list(cm.parameters()) + list(fm.parameters()), list(cm2.parameters()) + list(fm2.parameters())):
        assert torch.equal(p, q)
    # use_fine_model off: fine_model None in the file, the coarse model is aliased (train_nerf.py:91,94,186)
    checkpoint.save_nerf(path, 5, cm, cm, opt)
    ck = checkpoint.load(path)
    assert ck["fine_model"] is None
    c3, f3 = checkpoint.nerf_models(ck, dev())
    assert c3 is f3


@pytest.mark.parametrize("kind", ["nerf", "siren_nerf"])
def test_reference_layout_state_dict_into_fused_module(kind, tmp_path):
    """A checkpoint written by the reference holds state dicts of ITS classes: ModuleLists of Dense / Siren layers
    whose parameters are `linear.weight`-free plain `weight` / `bias` entries (nerf/nerf.py:5-28, 97-117).  Rebuilt
    here from nn.Linear pieces with those names, saved as the reference saves, loaded weights-only, rendered by both
    the look-alike (recognised by layout) and the fused module built from the file."""
    from mirender import checkpoint, fields, render_core
    k_in, k_skip, k_dir = (60, 316, 280) if kind == "nerf" else (3, 259, 259)

    class Dense(torch.nn.Linear):                       # carries its activation's name like nerf/nerf.py:15
        def __init__(self, i, o, activation="linear"):
            super().__init__(i, o)
            self.activation_name = activation

    class Siren(torch.nn.Linear):                       # nerf/nerf.py:97-117: the class IS the activation
        pass

    H = (lambda i, o: Dense(i, o, "relu")) if kind == "nerf" else Siren

    class RefLike(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers_pos = torch.nn.ModuleList([H(k_in, 256)] + [H(256, 256) for _ in range(4)] + [H(k_skip, 256), H(256, 256), H(256, 256)])
            self.layers_dir = torch.nn.ModuleList([Dense(256, 256), H(k_dir, 128)])
            self.output_layer_sigma = Dense(256, 1, "relu")
            self.output_layer_rgb = Dense(128, 3, "sigmoid")

    sd = synth.state_dict(kind, seed=72, sharp="medium", bias_jitter=0.05)
    ref_like = RefLike()
    ref_like.load_state_dict(sd)
    path = os.path.join(str(tmp_path), "000010.tar")
    torch.save({"global_step": 10, "coarse_model": ref_like.state_dict(), "fine_model": None,
                "optimizer": torch.optim.Adam(ref_like.parameters()).state_dict()}, path)
    cm, fm = checkpoint.nerf_models(checkpoint.load(path), dev())
    assert cm is fm and type(cm) is {"nerf": fields.NeRF, "siren_nerf": fields.SirenNeRF}[kind]
    ref_like = ref_like.to(dev())
    assert fields.as_packed_field(ref_like).kind == fields.as_packed_field(cm).kind
    rays, tr = _rays(), synth.t_rand(200, 16, seed=4).to(dev())
    with torch.no_grad():
        a = render_core.render_rays(rays, 2.0, 6.0, ref_like, ref_like, 16, 24, t_rand=tr)
        b = render_core.render_rays(rays, 2.0, 6.0, cm, fm, 16, 24, t_rand=tr)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_pigan_checkpoint_round_trip(tmp_path):
    from mirender import checkpoint, pigan
    torch.manual_seed(1)
    gen = pigan.Generator(32, 16, near=0.5, far=1.5, fov=12, coarse_samples=6, fine_samples=12).to(dev())
    gen.film_siren_nerf.load_state_dict(synth.state_dict("film_siren_nerf", seed=73, sharp="medium"))
    g_opt = torch.optim.Adam(gen.parameters(), lr=5e-5, betas=(0.0, 0.9))
    z = torch.randn(2, 32, device=dev())
    tr = synth.t_rand(2 * 16 * 16, 6, seed=5).to(dev())
    gen(z, [0.1, -0.2], [0.0, 0.1], t_rand=tr).square().mean().backward()
    g_opt.step()
    wrapped = torch.nn.DataParallel(gen, device_ids=[0])           # the script saves generator.module.state_dict()
    path = checkpoint.checkpoint_path(str(tmp_path), 42)
    checkpoint.save_pigan(path, 42, {"g_loss": [0.5], "d_loss": [1.5]}, wrapped, None, g_opt, None)
    ck = checkpoint.load(path)
    assert set(ck) == {"global_step", "loss_log", "generator", "discriminator", "g_optimizer", "d_optimizer"}
    assert ck["loss_log"] == {"g_loss": [0.5], "d_loss": [1.5]} and list(ck["generator"]) == list(gen.state_dict())
    gen2 = checkpoint.pigan_generator(ck, 16, dev(), near=0.5, far=1.5, fov=12, coarse_samples=6, fine_samples=12)
    g_opt2 = torch.optim.Adam(gen2.parameters(), lr=5e-5, betas=(0.0, 0.9))
    g_opt2.load_state_dict(ck["g_optimizer"])
    with torch.no_grad():
        a = gen(z, [0.1, -0.2], [0.0, 0.1], t_rand=tr)
        b = gen2(z, [0.1, -0.2], [0.0, 0.1], t_rand=tr)
    assert torch.equal(a, b)
    # Generator.render() after set_film_params, the demo / synthesis call (pi_GAN/modules.py:196-197, utils.py:198)
    film = gen2.get_mapping(z)
    gen.set_film_params(film[0].detach())
    gen2.set_film_params(film[0].detach())
    torch.manual_seed(7)
    ia = gen.render(0.2, -0.1)
    torch.manual_seed(7)
    ib = gen2.render(0.2, -0.1)
    assert tuple(ia.shape) == (16, 16, 3) and torch.equal(ia, ib)
    assert np.isfinite(ia.detach().cpu().numpy()).all()
