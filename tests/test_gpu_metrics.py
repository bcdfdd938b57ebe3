"""GPU parity of the evaluation stages through the C ABI:
frame metrics (mse / psnr / pytorch_ssim.ssim) against fixture F8 (generated from the reference's own
pytorch_ssim package) and the oracle; create_mesh's voxel-grid queries against the oracle restatement.

Tolerances: ssim 2e-5 absolute (the kernel applies the gaussian separably, the reference as one 121-tap 2-D
window: fp32 rounding only), mse 1e-7 relative, psnr 1e-4 dB (north_star asks for 0.05 dB); grid points
bit-exact; -sigma on the grid 1e-4 like every other field output."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fields as ofields, grid as G, metrics as M, synth  # noqa: E402

SSIM_TOL = 2e-5


def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def mi():
    from mirender import _lib, fields, grid, metrics
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    _lib.load()
    return type("MI", (), {"fields": fields, "grid": grid, "metrics": metrics, "lib": _lib})


@pytest.mark.parametrize("name", ["ragged", "frame", "tiny", "wide"])
def test_metrics_golden(mi, golden, name):
    g = golden("metrics_f8")
    a, b = torch.from_numpy(g[f"{name}.img1"]).to(dev()), torch.from_numpy(g[f"{name}.img2"]).to(dev())
    assert abs(float(mi.metrics.ssim(a, b)) - float(g[f"{name}.ssim"])) <= SSIM_TOL
    per = mi.metrics.ssim(a, b, size_average=False).cpu().numpy()
    assert np.abs(per - g[f"{name}.ssim_per_image"]).max() <= SSIM_TOL
    assert abs(float(mi.metrics.ssim(a, b, window_size=7)) - float(g[f"{name}.ssim_w7"])) <= SSIM_TOL
    assert abs(float(mi.metrics.SSIM()(a, b)) - float(g[f"{name}.ssim"])) <= SSIM_TOL
    assert abs(float(mi.metrics.mse(a, b)) / float(g[f"{name}.mse"]) - 1.0) <= 1e-6
    assert abs(float(mi.metrics.psnr(a, b)) - float(g[f"{name}.psnr"])) <= 1e-4


def test_metrics_degenerate(mi, golden):
    g = golden("metrics_f8")
    a = torch.from_numpy(g["same.img1"]).to(dev())
    assert abs(float(mi.metrics.ssim(a, a.clone())) - float(g["same.ssim"])) <= SSIM_TOL
    flat = torch.full((1, 3, 33, 33), 0.25, device=dev())
    assert abs(float(mi.metrics.ssim(flat, torch.full_like(flat, 0.5))) - float(g["flat.ssim_vs_half"])) <= SSIM_TOL
    assert float(mi.metrics.mse(a, a.clone())) == 0.0


def test_metrics_full_frame_vs_oracle_and_properties(mi):
    """800x800 (BASELINE frame size): against the oracle, plus symmetry and batch-independence."""
    gen = torch.Generator().manual_seed(4)
    a = torch.rand((2, 3, 800, 800), generator=gen)
    a = torch.nn.functional.avg_pool2d(a, 5, stride=1, padding=2)              # smooth: realistic local variance
    b = (a + 0.03 * torch.randn(a.shape, generator=gen)).clamp(0, 1)
    ad, bd = a.to(dev()), b.to(dev())
    got = mi.metrics.image_metrics(ad, bd).cpu()
    assert abs(float(got[:, 1].mean()) - float(M.ssim(a, b))) <= SSIM_TOL
    assert np.abs(got[:, 1].numpy() - M.ssim(a, b, size_average=False).numpy()).max() <= SSIM_TOL
    assert abs(float(got[:, 0].mean()) / float(M.mse(a, b)) - 1.0) <= 1e-6
    assert abs(float(mi.metrics.psnr(ad, bd)) - float(M.psnr(a, b))) <= 1e-4
    swapped = mi.metrics.image_metrics(bd, ad).cpu()
    assert torch.equal(swapped, got)                                           # every term is symmetric
    single = mi.metrics.image_metrics(ad[1:], bd[1:]).cpu()
    assert torch.equal(single[0], got[1])                                      # images do not interact


def test_metrics_rejects_bad_input(mi):
    a = torch.rand((1, 3, 8, 8), device=dev())
    with pytest.raises(mi.lib.MiRenderError):
        mi.metrics.image_metrics(a, a[:, :2])
    with pytest.raises(mi.lib.MiRenderError):
        mi.metrics.image_metrics(a.cpu(), a.cpu())
    with pytest.raises(mi.lib.MiRenderError):
        mi.metrics.ssim(a, a, window_size=10)


@pytest.mark.parametrize("n,head,count", [(5, 0, 125), (17, 100, 3000), (64, 64 ** 3 - 999, 999), (256, 256 ** 3 - 70000, 70000)])
def test_grid_points_bit_exact(mi, n, head, count):
    got = mi.grid.grid_points(n, (-0.1, -0.1, -0.1), 0.2 / (n - 1), head, count, dev()).cpu().numpy()
    if n <= 64:
        exp = G.grid_samples(n).numpy()[head:head + count]
    else:   # the full 256^3 table is 268 MB on the host: restate the three lines for the requested rows only
        idx = torch.arange(head, head + count, dtype=torch.long)
        vs, o = 0.2 / (n - 1), -0.1
        exp = torch.stack([(torch.floor_divide(torch.floor_divide(idx, n), n) % n).float() * vs + o,
                           (torch.floor_divide(idx, n) % n).float() * vs + o, (idx % n).float() * vs + o], -1).numpy()
    assert np.array_equal(got[:, :3], exp)
    assert not got[:, 3:].any()


def test_grid_points_axis_order(mi):
    got = mi.grid.grid_points(3, (1.0, 2.0, 3.0), 0.5, 0, 27, dev()).cpu().numpy()
    assert got[0, :3].tolist() == [3.0, 2.0, 1.0] and got[-1, :3].tolist() == [4.0, 3.0, 2.0]
    with pytest.raises(mi.lib.MiRenderError):
        mi.grid.grid_points(3, (0, 0, 0), 0.5, 20, 8, dev())


@pytest.mark.parametrize("kind", ["film_siren_nerf", "nerf"])
def test_density_grid_vs_oracle(mi, kind):
    n = 12
    sd = synth.state_dict(kind, seed=21)
    film = synth.film_params(1, seed=3) if kind.startswith("film") else None
    oracle_field = ofields.make_field(kind, sd, None if film is None else film[0])
    model = mi.fields.field_from_state_dict(sd, dev())
    got = mi.grid.density_grid(model, n=n, max_batch=500, film=None if film is None else film.to(dev())).cpu()
    exp = G.density_grid(oracle_field, n, max_batch=500)
    assert got.shape == (n, n, n)
    assert float((got - exp).abs().max()) <= 1e-4
