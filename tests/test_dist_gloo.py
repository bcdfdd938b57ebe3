"""CPU, world_size 2 (gloo): the ray-shard + all-gather frame assembly and the gradient all-reduce.

The shard renderer here is the CPU oracle (tests may use it as the checker); on the GPU the same
`render_image_sharded` is driven by the HIP path with backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fields as ofields, render_ref as R, synth

W, H, NC, NF = 13, 9, 8, 8         # 117 rays: odd, so the two shards differ in size


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_shard_renderer(w=W, h=H):
    sd = synth.state_dict("tiny_nerf", seed=4, sharp=True)
    f = ofields.make_field("tiny_nerf", sd)
    pose = synth.pose_degrees(4.0, 20.0, -30.0)
    rays = torch.from_numpy(R.rays_from_camera(w, h, 1.3875 * w, pose))
    tr = synth.t_rand(w * h, NC, seed=2)

    def shard(ray0, n):
        if n == 0:                                   # more ranks than rays: this rank renders nothing
            return torch.zeros(0, 3), torch.zeros(0), torch.zeros(0)
        with torch.no_grad():
            t = R.render_rays(rays[ray0:ray0 + n], 2.0, 6.0, f, f, NC, NF, tr[ray0:ray0 + n])
        return t.rgb_f, t.depth_f, t.acc_f
    return shard


def _worker(rank, world, port, out, w=W, h=H):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirender import dist as mdist
    torch.set_num_threads(2 if world <= 2 else 1)
    rgb, depth, acc = mdist.render_image_sharded(_oracle_shard_renderer(w, h), w, h)
    # gradient all-reduce: rank r holds grad = r+1 everywhere -> mean 1.5
    p = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
    for q in p:
        q.grad = torch.full_like(q, float(rank + 1))
    # a parameter only the even ranks' shards reached (an empty shard, a loss that touched one model only): it still takes
    # part on every rank, as zeros where it is missing - the flat buffers must have ONE layout or the collective hangs
    p.append(torch.nn.Parameter(torch.zeros(4)))
    if rank % 2 == 0:
        p[-1].grad = torch.full((4,), float(rank + 1))
    mdist.allreduce_grads(p)
    assert p[-1].grad is not None
    out[rank] = (rgb.numpy(), depth.numpy(), acc.numpy(), [q.grad.clone().numpy() for q in p])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_frame_equals_single_process():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ref = _oracle_shard_renderer()(0, W * H)
    for rank in range(world):
        rgb, depth, acc, grads = out[rank]
        assert rgb.shape == (H, W, 3) and depth.shape == (H, W, 1) and acc.shape == (H, W, 1)
        assert np.array_equal(rgb.reshape(-1, 3), ref[0].numpy())     # shards are independent: exact
        assert np.array_equal(depth.reshape(-1), ref[1].numpy()) and np.array_equal(acc.reshape(-1), ref[2].numpy())
        assert all(np.allclose(g, 1.5) for g in grads[:2]) and np.allclose(grads[2], 0.5)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("w,h", [(13, 9), (3, 2)])
def test_eight_ranks_ragged_and_empty_shards(w, h):
    """world_size 8, the rank count of the driver's scaling run: 117 rays split 15/15/15/15/15/14/14/14 (ragged: the shards
    are padded to the longest for the one all-gather and cut back afterwards), and 6 rays over 8 ranks (two ranks render
    nothing and still take part in the collective).  Every rank ends with the single-process frame, bit for bit, and the
    mean of eight ranks' gradients."""
    from mirender import dist as mdist
    world = 8
    sizes = [b - a for a, b in (mdist.shard_range(w * h, r, world) for r in range(world))]
    assert sum(sizes) == w * h and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, w, h), nprocs=world, join=True)
    # the CPU oracle's own result depends a little on how many rays go through MKL at once (a 1-ray sgemm rounds differently
    # from a 15-ray one, and the sharp head's resampling amplifies it): the reference frame is assembled from the same
    # shard sizes in ONE process, which is exactly what the collective has to reproduce; the whole-frame render agrees to 1e-2
    one = _oracle_shard_renderer(w, h)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                        # like the eight workers (MKL's result also depends on its thread count)
    try:
        shards = [one(*(lambda a, b: (a, b - a))(*mdist.shard_range(w * h, r, world))) for r in range(world)]
        whole = one(0, w * h)
    finally:
        torch.set_num_threads(threads)
    ref = [torch.cat([s[k] for s in shards]).numpy() for k in range(3)]
    assert np.abs(ref[0] - whole[0].numpy()).max() < 1e-2
    for rank in range(world):
        rgb, depth, acc, grads = out[rank]
        assert rgb.shape == (h, w, 3) and depth.shape == (h, w, 1) and acc.shape == (h, w, 1)
        assert np.array_equal(rgb.reshape(-1, 3), ref[0])
        assert np.array_equal(depth.reshape(-1), ref[1]) and np.array_equal(acc.reshape(-1), ref[2])
        assert all(np.allclose(g, 4.5) for g in grads[:2])             # mean of 1..8
        assert np.allclose(grads[2], 2.0)                               # (1 + 3 + 5 + 7) / 8: the odd ranks sent zeros


def _one_rank_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirender import dist as mdist
    calls = {"gather": 0, "reduce": 0}
    real_g, real_r = dist.all_gather_into_tensor, dist.all_reduce

    def g(*a, **k):
        calls["gather"] += 1
        return real_g(*a, **k)

    def r(*a, **k):
        calls["reduce"] += 1
        return real_r(*a, **k)
    dist.all_gather_into_tensor, dist.all_reduce = g, r
    res = {}
    for force in (False, True):
        mdist.FORCE_COLLECTIVE = force
        rgb, depth, acc = mdist.render_image_sharded(_oracle_shard_renderer(), W, H)
        p = [torch.nn.Parameter(torch.zeros(5, 3))]
        p[0].grad = torch.full((5, 3), 2.5)
        mdist.allreduce_grads(p)
        res[force] = (rgb.numpy(), p[0].grad.numpy().copy(), dict(calls))
    out[0] = res
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_one_rank_group_skips_its_collectives_unless_forced():
    """With one rank there is nothing to exchange: render_image_sharded / allreduce_grads skip the collectives - unless
    mirender.dist.FORCE_COLLECTIVE is on (how the RCCL calls are executed on a one-GPU box, tests/test_gpu_rccl.py).
    Either way the frame and the gradients are unchanged."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_one_rank_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    res = out[0]
    assert res[False][2] == {"gather": 0, "reduce": 0}
    assert res[True][2] == {"gather": 1, "reduce": 1}
    assert np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1])
    assert np.allclose(res[True][1], 2.5)


def test_bench_finds_the_newest_rounds_profile(tmp_path, monkeypatch):
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    for name in ("r01_pmc_c4.json", "r03_pmc_c4.json", "r02_pmc_c4.json", "r02_pmc_nerf_fwd.json"):
        (prof / name).write_text("{}")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert os.path.basename(bench.newest_profile("pmc_c4.json")) == "r03_pmc_c4.json"
    assert os.path.basename(bench.newest_profile("pmc_nerf_fwd.json")) == "r02_pmc_nerf_fwd.json"
    assert bench.newest_profile("pmc_c5.json") is None
