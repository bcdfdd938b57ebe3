"""GPU parity of the nerf training-loop data path through the C ABI:
the fused loss + gradient seed (nerf/train_nerf.py:158-167) against fixture F7 and the oracle's autograd, the
device-built rays_rgba table (train_nerf.py:64-68, 78-82) bit-exact against the oracle, batching semantics, and
one full step (RayBank -> render_rays -> nerf_loss -> backward) against the same step with the loss in torch ops."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import parity, render_ref as R, synth, train_ref as T  # noqa: E402


def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def mi():
    from mirender import _lib, fields, ops, render_core, train
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    _lib.load()
    return type("MI", (), {"fields": fields, "train": train, "render_core": render_core, "lib": _lib, "ops": ops})


def test_loss_golden_f7(mi, golden):
    g = golden("nerf_grad_f7")
    t = lambda k: torch.from_numpy(g[k]).to(dev())  # noqa: E731
    tgt = t("target")
    outs = (t("rgb_c"), None, t("acc_c"), t("rgb_f"), None, t("acc_f"))
    loss, psnr = mi.train.nerf_loss(outs, tgt[:, :3], tgt[:, 3], use_alpha=True, use_fine_model=True)
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    assert abs(float(psnr) + 10 * np.log10(float(((g["rgb_f"] - g["target"][:, :3]) ** 2).mean()))) <= 1e-4


@pytest.mark.parametrize("n", [1, 37, 256, 1024, 4099])
@pytest.mark.parametrize("use_alpha,use_fine", [(False, True), (True, True), (True, False), (False, False)])
def test_loss_and_seed_vs_oracle_autograd(mi, n, use_alpha, use_fine):
    gen = torch.Generator().manual_seed(n)
    cpu = [torch.rand(s, generator=gen).requires_grad_(True) for s in ((n, 3), (n,), (n, 3), (n,))]
    tgt = torch.rand((n, 4), generator=gen)
    ref_loss, ref_psnr = T.nerf_loss((cpu[0], None, cpu[1], cpu[2], None, cpu[3]), tgt[:, :3], tgt[:, 3], use_alpha, use_fine)
    (ref_loss * 1.7).backward()
    gpu = [c.detach().to(dev()).requires_grad_(True) for c in cpu]
    loss, psnr = mi.train.nerf_loss((gpu[0], None, gpu[1], gpu[2], None, gpu[3]), tgt[:, :3].to(dev()), tgt[:, 3].to(dev()),
                                    use_alpha, use_fine)
    (loss * 1.7).backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-7 * max(1.0, abs(float(ref_loss.detach())))
    assert abs(float(psnr.detach()) - float(ref_psnr.detach())) <= 1e-4
    worst = 0.0
    for a, b in zip(gpu, cpu):
        want = torch.zeros_like(b) if b.grad is None else b.grad
        worst = max(worst, float((a.grad.cpu() - want).abs().max()))
    parity.record(case=f"nerf_loss seeds n={n} use_alpha={use_alpha} use_fine={use_fine}", stage="gradient (loss seeds)",
                  qty="d loss / d (rgb_c, acc_c, rgb_f, acc_f)", err_vs_oracle32=worst, tol=1e-7, unit="max abs",
                  reference="oracle fp32 autograd (train_nerf.py:158-167)", active="hard", passed=worst <= 1e-7)
    assert worst <= 1e-7


def test_ray_bank_bit_exact(mi):
    W, H, n = 8, 6, 3
    focal = 1.3875 * W
    poses = np.stack([synth.pose_degrees(4.0, th, -30.0) for th in (37.0, -120.0, 5.0)]).astype(np.float32)
    imgs = np.random.Generator(np.random.PCG64(3)).random((n, H, W, 4), dtype=np.float32)
    for white in (True, False):
        bank = mi.train.RayBank(imgs, poses, focal, dev(), white_bkgd=white)
        exp = T.rays_rgba(imgs, poses, W, H, focal, white_bkgd=white)
        assert np.array_equal(bank.table.cpu().numpy(), exp)


def test_ray_bank_batches_and_epoch_reshuffle(mi):
    W, H, n = 10, 7, 2
    poses = np.stack([synth.pose_degrees(4.0, th, -30.0) for th in (0.0, 90.0)]).astype(np.float32)
    imgs = np.random.Generator(np.random.PCG64(4)).random((n, H, W, 4), dtype=np.float32)
    bank = mi.train.RayBank(imgs, poses, 1.3875 * W, dev())
    ref = {tuple(r) for r in bank.table.cpu().numpy().round(6).tolist()}
    bank.shuffle()
    first = bank.table.clone()
    bs = 32
    seen = []
    for _ in range(-(-len(bank) // bs)):                      # one epoch: ceil(140/32) = 5 batches, the last ragged
        rays, rgb, alpha = bank.batch(bs)
        assert rays.shape[1:] == (2, 3) and rgb.shape[1] == 3 and alpha.dim() == 1 and rays.shape[0] == rgb.shape[0] == alpha.shape[0]
        seen.append(torch.cat([rays.reshape(-1, 6), rgb, alpha[:, None]], 1))
    seen = torch.cat(seen)
    assert seen.shape == first.shape and torch.equal(seen, first)          # the epoch walks the shuffled table once
    assert {tuple(r) for r in seen.cpu().numpy().round(6).tolist()} == ref  # ... which is a permutation of the rows
    assert bank.batch_idx == 0 and not torch.equal(bank.table, first)      # and the next epoch is reshuffled


def test_one_training_step_matches_torch_loss(mi):
    """RayBank -> render_rays -> fused loss -> backward == the same step with train_nerf.py:158-167 in torch ops."""
    W, H = 12, 9
    poses = np.stack([synth.pose_degrees(4.0, th, -30.0) for th in (20.0, -60.0)]).astype(np.float32)
    imgs = np.random.Generator(np.random.PCG64(5)).random((2, H, W, 4), dtype=np.float32)
    bank = mi.train.RayBank(imgs, poses, 1.3875 * W, dev(), generator=torch.Generator(device=dev()).manual_seed(1))
    bank.shuffle()
    rays, rgb, alpha = bank.batch(64)
    cm = mi.fields.field_from_state_dict(synth.state_dict("nerf", seed=60, sharp=True, bias_jitter=0.05), dev())
    fm = mi.fields.field_from_state_dict(synth.state_dict("nerf", seed=61, sharp=True, bias_jitter=0.05), dev())
    params = list(cm.parameters()) + list(fm.parameters())
    tr = synth.t_rand(64, 16, seed=2).to(dev())
    grads = []
    for fused in (True, False):
        for p in params:
            p.grad = None
        outs = mi.render_core.render_rays(rays, 2.0, 6.0, cm, fm, 16, 24, t_rand=tr)
        if fused:
            loss, psnr = mi.train.nerf_loss(outs, rgb, alpha, use_alpha=True, use_fine_model=True)
        else:
            loss, psnr = T.nerf_loss(outs, rgb, alpha, use_alpha=True, use_fine_model=True)
        loss.backward()
        grads.append((float(loss.detach()), float(psnr.detach()), [p.grad.clone() for p in params]))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 and abs(grads[0][1] - grads[1][1]) <= 1e-4
    names = [f"{w}.{k}" for w, m_ in (("coarse", cm), ("fine", fm)) for k, _ in m_.named_parameters()]
    for name, a, b in zip(names, grads[0][2], grads[1][2]):
        parity.gate_grad("one nerf step 64 rays 16+24: fused loss kernel vs train_nerf.py:158-167 in torch ops (same HIP backward)",
                         name, a.cpu(), b.cpu(), tol=1e-5, stage="gradient (loss path)")
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
    assert mi.train.decayed_lr(5e-4, 500, 250000) == 5e-4 * 0.1 ** 0.5


@pytest.mark.parametrize("kinds", [("nerf", "nerf"), ("tiny_nerf",), ("siren_nerf", "nerf"), ("film_siren_nerf",),
                                   ("film_siren_nerf_nodir", "tiny_nerf")])
def test_fused_adam_matches_torch_adam_and_keeps_the_streams_current(mi, kinds):
    """FusedAdam (mi_adam_step: Adam on every tensor + scatter of the new values into both packed MFMA streams, one
    launch) against torch.optim.Adam(lr, betas=(0.9, 0.999)) of nerf/train_nerf.py:98 on identical gradients for 10
    steps with the script's learning-rate decay (train_nerf.py:170-175): parameters and both Adam moments within 1e-7;
    after every step the streams the kernel patched in place equal a fresh pack of the updated parameters bit for
    bit (forward order and, where it exists, the transposed order of the backward chain)."""
    torch.manual_seed(0)
    mods_a = [mi.fields.field_from_state_dict(synth.state_dict(k, seed=80 + i, sharp="medium", bias_jitter=0.05), dev())
              for i, k in enumerate(kinds)]
    mods_b = [mi.fields.field_from_state_dict(synth.state_dict(k, seed=80 + i, sharp="medium", bias_jitter=0.05), dev())
              for i, k in enumerate(kinds)]
    pfs = [mi.fields.as_packed_field(m) for m in mods_a]
    pfs[0].refresh_bwd()                                   # field 0 has a transposed stream, field 1 (if any) not yet
    for pf in pfs:
        pf.refresh()
    fused = mi.train.FusedAdam(mods_a, lr=5e-4, betas=(0.9, 0.999))
    params_b = [p for m in mods_b for p in m.parameters()]
    ref = torch.optim.Adam(params_b, lr=5e-4, betas=(0.9, 0.999))
    assert [tuple(p.shape) for p in fused.params] == [tuple(p.shape) for p in params_b]
    gen = torch.Generator(device=dev()).manual_seed(1)
    for step in range(10):
        for pa, pb in zip(fused.params, params_b):
            g = torch.randn(pa.shape, device=dev(), generator=gen) * (10.0 ** float(torch.randint(-4, 1, (1,)).item()))
            pa.grad, pb.grad = g.clone(), g.clone()
        fused.step()
        ref.step()
        lr = mi.train.decayed_lr(5e-4, 0.5, step + 1)     # fast decay so the schedule matters within 10 steps
        for group in list(fused.param_groups) + list(ref.param_groups):
            group["lr"] = lr
        for pa, pb in zip(fused.params, params_b):
            assert float((pa - pb).abs().max()) <= 1e-7
        for pf in pfs:                                      # streams patched in place == repacked from scratch
            fresh = mi.fields.PackedField(pf.kind, pf.params)
            assert torch.equal(pf.packed, fresh.refresh())
            if pf.packed_bwd is not None:
                assert torch.equal(pf.packed_bwd, fresh.refresh_bwd())
    for pa, pb in zip(fused.params, params_b):
        sa, sb = fused.state[pa], ref.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == 10.0
        assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 1e-7 * max(1.0, float(sb["exp_avg"].abs().max()))
        assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 1e-7 * max(1.0, float(sb["exp_avg_sq"].abs().max()))
    # the renderer sees the updated weights without any repack (version counters did not move)
    x = torch.rand(300, 6, device=dev()) * 2 - 1
    film = synth.film_params(1, seed=2).to(dev())
    for ma, mb, k in zip(mods_a, mods_b, kinds):
        f = film if k.startswith("film") else None
        a = mi.fields.eval_points(mi.fields.as_packed_field(ma), x, f)
        b = mi.fields.eval_points(mi.fields.as_packed_field(mb), x, f)
        assert float((a - b).abs().max()) <= 1e-4
    # state dicts are interchangeable with torch.optim.Adam's in both directions (checkpoint 'optimizer' entry)
    sd = fused.state_dict()
    ref2 = torch.optim.Adam([p for m in mods_a for p in m.parameters()], lr=1.0)
    ref2.load_state_dict(sd)
    assert ref2.param_groups[0]["lr"] == fused.param_groups[0]["lr"] and float(ref2.state[fused.params[0]]["step"]) == 10.0
    fused2 = mi.train.FusedAdam(mods_b, lr=1.0)
    fused2.load_state_dict(ref.state_dict())
    for pa, pb in zip(fused.params, fused2.params):
        g = torch.randn(pa.shape, device=dev(), generator=gen) * 1e-2
        pa.grad, pb.grad = g.clone(), g.clone()
    fused.step()
    fused2.step()
    for pa, pb in zip(fused.params, fused2.params):
        assert float((pa - pb).abs().max()) <= 2e-7


def test_fused_adam_in_the_training_step(mi):
    """The loop of nerf/train_nerf.py:124-176 with FusedAdam in place of torch.optim.Adam: same losses (1e-5 relative
    after 6 steps); coarse_model is fine_model (use_fine_model off) is one field, updated once."""
    W, H = 12, 9
    poses = np.stack([synth.pose_degrees(4.0, th, -30.0) for th in (20.0, -60.0)]).astype(np.float32)
    imgs = np.random.Generator(np.random.PCG64(5)).random((2, H, W, 4), dtype=np.float32)
    losses = {}
    for which in ("fused", "torch"):
        bank = mi.train.RayBank(imgs, poses, 1.3875 * W, dev(), generator=torch.Generator(device=dev()).manual_seed(1))
        bank.shuffle()
        cm = mi.fields.field_from_state_dict(synth.state_dict("tiny_nerf", seed=60, sharp="medium", bias_jitter=0.05), dev())
        fm = mi.fields.field_from_state_dict(synth.state_dict("tiny_nerf", seed=61, sharp="medium", bias_jitter=0.05), dev())
        params = list(cm.parameters()) + list(fm.parameters())
        opt = mi.train.FusedAdam([cm, fm], lr=5e-4) if which == "fused" else torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
        out = []
        for step in range(6):
            rays, rgb, alpha = bank.batch(54)                 # 216 rays: four full batches per epoch
            outs = mi.render_core.render_rays(rays, 2.0, 6.0, cm, fm, 8, 8, t_rand=synth.t_rand(54, 8, seed=step).to(dev()))
            loss, _ = mi.train.nerf_loss(outs, rgb, alpha, use_alpha=True, use_fine_model=True)
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        losses[which] = out
    for a, b in zip(losses["fused"], losses["torch"]):
        assert abs(a - b) <= 1e-5 * abs(b), (losses,)
    assert losses["fused"][-1] < losses["fused"][0]
    m = mi.fields.field_from_state_dict(synth.state_dict("tiny_nerf", seed=62), dev())
    assert len(mi.train.FusedAdam([m, m]).fields) == 1
    # options plain Adam does not have are refused, not ignored
    opt = mi.train.FusedAdam(m)
    opt.param_groups[0]["weight_decay"] = 1e-2
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(mi.lib.MiRenderError):
        opt.step()


def test_ray_bank_float64_focal_and_reference_reshuffle_quirk(mi):
    """focal as nerf/data_loader.py:151 returns it (an np.float64 scalar): NumPy >= 2 then evaluates get_rays in fp64
    (train_nerf.py:78) before train_nerf.py:84 rounds to fp32 - the table must follow.  reshuffle=False reproduces the
    reference's dead epoch shuffle (train_nerf.py:144 assigns to a misspelt name): every epoch replays the first."""
    W, H, n = 9, 7, 2
    focal64 = np.float64(0.5 * W / np.tan(0.5 * 0.6911112))          # data_loader.py:151
    poses = np.stack([synth.pose_degrees(4.0, th, -30.0) for th in (37.0, -120.0)]).astype(np.float32)
    imgs = np.random.Generator(np.random.PCG64(3)).random((n, H, W, 4), dtype=np.float32)
    bank = mi.train.RayBank(imgs, poses, focal64, dev())
    exp = T.rays_rgba(imgs, poses, W, H, focal64)
    assert np.array_equal(bank.table.cpu().numpy(), exp)
    bank32 = mi.train.RayBank(imgs, poses, float(focal64), dev())
    assert np.array_equal(bank32.table.cpu().numpy(), T.rays_rgba(imgs, poses, W, H, float(focal64)))
    assert not np.array_equal(bank.table.cpu().numpy(), bank32.table.cpu().numpy())      # the promotion matters
    quirk = mi.train.RayBank(imgs, poses, float(focal64), dev(), reshuffle=False,
                             generator=torch.Generator(device=dev()).manual_seed(2))
    quirk.shuffle()
    first = [quirk.batch(50)[0].clone() for _ in range(3)]          # 126 rays: 3 batches per epoch
    again = [quirk.batch(50)[0].clone() for _ in range(3)]
    for a, b in zip(first, again):
        assert torch.equal(a, b)


def test_fused_adam_step_between_forward_and_backward_raises(mi):
    """FusedAdam rewrites the parameters from a raw kernel, so torch's version counters do not move; the
    field's own epoch does, and a forward still waiting for its backward must see it (mixed-weights gradients would be
    silent otherwise).  The streams themselves stay valid: the next forward needs no repack."""
    m = mi.fields.field_from_state_dict(synth.state_dict("tiny_nerf", seed=63, sharp="medium", bias_jitter=0.05), dev())
    pf = mi.fields.as_packed_field(m)
    rays = torch.from_numpy(R.rays_from_camera(12, 12, 16.0, synth.pose_degrees(4.0, 5.0, -30.0))).to(dev())
    tr = synth.t_rand(144, 8, seed=1).to(dev())
    opt = mi.train.FusedAdam(m, lr=1e-3)
    out = mi.render_core.render_rays(rays, 2.0, 6.0, m, m, 8, 8, t_rand=tr)
    out[3].square().mean().backward()                     # gradients for the step
    stale = mi.render_core.render_rays(rays, 2.0, 6.0, m, m, 8, 8, t_rand=tr)
    stamp = pf._versions
    opt.step()
    assert pf._versions != stamp and pf._versions == pf.versions()        # re-stamped: still current, no repack due
    with pytest.raises(RuntimeError, match="modified in place"):
        stale[3].square().mean().backward()
    packed_before = pf.packed.clone()
    fresh = mi.render_core.render_rays(rays, 2.0, 6.0, m, m, 8, 8, t_rand=tr)
    assert torch.equal(pf.packed, packed_before)           # the forward after the step used the patched stream as is
    fresh[3].square().mean().backward()                    # and its own backward is fine


def test_adam_step_rejects_tensors_that_are_not_weight_bias_pairs(mi):
    """mi_adam_step takes each weight's row length from the bias that follows it: a zero-sized or non-dividing bias is
    an argument error (MI_EINVAL), not a division fault."""
    import ctypes
    m = mi.fields.field_from_state_dict(synth.state_dict("tiny_nerf", seed=64), dev())
    pf = mi.fields.as_packed_field(m)
    ps = list(pf.params)
    z = [torch.zeros_like(p) for p in ps]
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])  # noqa: E731
    kinds = (ctypes.c_int * 1)(pf.kind)
    for bad_index, bad_value in ((1, 0), (1, 7)):
        numel = [p.numel() for p in ps]
        numel[bad_index] = bad_value
        rc = mi.lib.load().mi_adam_step(1, kinds, arr(ps), arr(z), arr(z), arr(z), (ctypes.c_int64 * len(ps))(*numel),
                                        -1e-3, 0.1, 0.999, 0.001, 1e-8, 1.0, arr([pf.refresh()]),
                                        (ctypes.c_void_p * 1)(None), mi.lib.stream_ptr(dev()))
        assert rc == -1 and b"weight" in mi.lib.load().mi_last_error()


def test_workspace_release_accepts_any_spelling_of_the_device(mi):
    m = mi.fields.field_from_state_dict(synth.state_dict("tiny_nerf", seed=65), dev())
    rays = torch.from_numpy(R.rays_from_camera(8, 8, 11.0, synth.pose_degrees(4.0, 5.0, -30.0))).to(dev())
    for spelling in ("cuda", "cuda:0", torch.device("cuda"), torch.device("cuda", 0), 0, None):
        with torch.no_grad():
            mi.render_core.render_rays(rays, 2.0, 6.0, m, m, 8, 8, seed=1)
        assert any(k[0] == "cuda:0" for k in mi.ops._Workspace.bufs)
        mi.ops._Workspace.release(spelling)
        assert not any(k[0] == "cuda:0" for k in mi.ops._Workspace.bufs), spelling
