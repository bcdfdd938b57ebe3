"""GPU: the render path inside a HIP graph.

include/mi_render.h promises "no allocation, no synchronisation inside: callers pass outputs and workspace; all
launches are asynchronous on `stream`".  The strictest check of that promise is stream capture: a captured stream
refuses synchronous calls, host-blocking copies and launches on other streams.  `render_rays_fused` (six launches, or
eight on the one-field path) is captured into a `torch.cuda.CUDAGraph` after one warm-up call (which packs the weight
streams and fills the linspace tables - the only host-to-device traffic of the path), replayed, replayed again with new
contents in the captured input buffers, and must equal the eager call bit for bit each time.  What a caller that
renders many small frames (render_video at 100x100) uses to take the launch gaps out."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import render_ref as R, synth  # noqa: E402


def dev():
    return torch.device("cuda", 0)


def _field(kind, seed):
    from mirender import fields
    m = fields.field_from_state_dict(synth.state_dict(kind, seed=seed, sharp="medium", bias_jitter=0.05), dev())
    return fields.as_packed_field(m)


@pytest.mark.parametrize("kind,shared,n,nc,nf,seeded", [("nerf", False, 1000, 32, 64, False), ("tiny_nerf", True, 2500, 32, 0, True),
                                                       ("film_siren_nerf", True, 512, 12, 24, False),
                                                       ("siren_nerf", True, 257, 8, 16, True)])
def test_render_rays_captured_in_a_graph_equals_eager(kind, shared, n, nc, nf, seeded):
    from mirender import ops
    is_film = kind.startswith("film")
    pf_c = _field(kind, 3)
    pf_f = pf_c if shared else _field(kind, 4)
    film = synth.film_params(2, seed=1).to(dev()) if is_film else None
    near, far = (0.5, 1.5) if is_film else (2.0, 6.0)

    def inputs(k):
        pose = synth.pose_radians(1.0, 0.1 * k, -0.1) if is_film else synth.pose_degrees(4.0, 20.0 + 40 * k, -30.0)
        rays = torch.from_numpy(R.rays_from_camera(60, 60, 180.0 if is_film else 83.0, pose)[:n]).to(dev())
        return rays, (None if seeded else synth.t_rand(n, nc, seed=10 + k).to(dev()))

    rays0, tr0 = inputs(0)
    s_rays = rays0.clone()
    s_tr = None if tr0 is None else tr0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():                  # warm-up on the capture stream's side: packs, tables
        ops.render_rays_fused(pf_c, pf_f, s_rays, near, far, nc, nf, film, s_tr, seed=5)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        s_out = ops.render_rays_fused(pf_c, pf_f, s_rays, near, far, nc, nf, film, s_tr, seed=5)
    for k in (0, 1, 2):
        rays, tr = inputs(k)
        s_rays.copy_(rays)
        if tr is not None:
            s_tr.copy_(tr)
        graph.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in s_out]
        with torch.no_grad():
            want = ops.render_rays_fused(pf_c, pf_f, rays, near, far, nc, nf, film, tr, seed=5)
        for a, b in zip(got, want):
            assert torch.equal(a, b), (kind, k)
    assert float(got[3].std()) > 1e-3
