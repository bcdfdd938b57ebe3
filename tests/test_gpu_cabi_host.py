"""GPU: the C ABI used from a program with no Python and no torch in it.

`tests/cabi/cabi_host.cpp` (built by `__graft_entry__.build()` / `csrc/build.py` with the link line a C or C++ host
uses: `-I include -lmirender`, HIP from /opt/rocm) allocates with hipMalloc, asks the library for the parameter shapes
(`mi_field_param_shape`), fills weights from a splitmix64 stream, and makes the calls of the reference's `render_image`
(nerf/render.py:150-167): `mi_gen_rays`, `mi_field_pack`, `mi_render_workspace_bytes`, `mi_render_rays` on its own
hipStream_t.  Here the same stream of numbers is produced in NumPy, the same calls go through the ctypes binding on
torch's allocator and torch's stream, and the six outputs must agree bit for bit: the boundary carries no hidden torch
state (allocator alignment, current stream, device guard)."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "cabi", "cabi_host")
MASK = (1 << 64) - 1


def splitmix_uniform(seed, count, bound):
    """cabi_host.cpp's Rng: `count` draws of float32(u * 2 - 1) * float32(bound), u = (next() >> 11) / 2^53."""
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, count + 1, dtype=np.uint64)
        z = (s ^ (s >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0
    return u.astype(np.float32) * np.float32(bound), (int(seed) + 0x9E3779B97F4A7C15 * count) & MASK


def host_params(kind, model_index, dev):
    from mirender import fields
    seed, params = 1000 + model_index, []
    relu = kind in (fields.NERF, fields.TINY_NERF)
    for _, (o, i) in fields.SPECS[kind]:
        sin_layer = (not relu) and o not in (1, 3)
        bound = np.sqrt(np.float32(6.0) / np.float32(i)) * np.float32((0.25 if i <= 3 else 0.03125) if sin_layer else 0.875)
        w, seed = splitmix_uniform(seed, o * i, bound)
        b, seed = splitmix_uniform(seed, o, 0.05)
        params += [torch.from_numpy(w.reshape(o, i)).to(dev), torch.from_numpy(b).to(dev)]
    return params


@pytest.mark.parametrize("kind,w,h,nc,nf,shared", [(0, 40, 25, 16, 32, 0), (0, 33, 20, 8, 8, 1), (1, 32, 16, 8, 16, 0),
                                                  (2, 48, 32, 12, 24, 1), (3, 20, 20, 6, 12, 1), (4, 50, 50, 32, 0, 0),
                                                  (2, 31, 17, 12, 24, 0)])
def test_cpp_host_equals_ctypes_host(kind, w, h, nc, nf, shared, tmp_path):
    from mirender import fields, ops
    assert os.path.exists(HOST), f"{HOST} missing: run python msra-practice-project_amd/csrc/build.py"
    out = tmp_path / "out.bin"
    run = subprocess.run([HOST, *map(str, (kind, w, h, nc, nf, shared)), str(out)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    n = w * h
    got = np.fromfile(out, dtype=np.float32)
    assert got.size == 10 * n
    dev = torch.device("cuda", 0)
    pf_c = fields.PackedField(kind, host_params(kind, 0, dev))
    pf_f = pf_c if shared else fields.PackedField(kind, host_params(kind, 1, dev))
    is_film = kind in (fields.FILM_SIREN_NERF, fields.FILM_SIREN_NERF_NODIR)
    film = None
    if is_film:
        f, _ = splitmix_uniform(77, 9 * 512, 0.25)
        base = np.tile(np.concatenate([np.ones(256, np.float32), np.zeros(256, np.float32)]), 9)
        film = torch.from_numpy((base + f).reshape(1, 9, 512)).to(dev)
    r = 1.0 if is_film else 4.0
    c2w = np.array([[0.96, 0.0, 0.28, 0.3], [0.0, 1.0, 0.0, -0.2], [-0.28, 0.0, 0.96, r]], np.float32)
    focal = w / 2.0 / 0.10510423526567646 if is_film else 1.3875 * w
    near, far = (0.5, 1.5) if is_film else (2.0, 6.0)
    rays = ops.gen_rays(w, h, float(focal), c2w, dev)
    with torch.no_grad():
        outs = ops.render_rays_fused(pf_c, pf_f, rays, near, far, nc, nf, film, None, seed=42, exact_linspace=False)
    torch.cuda.synchronize()
    want = np.concatenate([o.cpu().numpy().ravel() for o in outs])
    assert np.isfinite(want).all() and float(want[5 * n:8 * n].std()) > 1e-3            # a picture, not a constant
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        f"C++ host and ctypes host differ in {int((got != want).sum())} of {got.size} values"
