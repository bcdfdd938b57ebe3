"""GPU, world_size 2 on ONE device (gloo transport, HIP render path): `render_image_dist` - the function
bench.py runs under torch.distributed.run with backend nccl (= RCCL) on N GPUs - and the data-parallel training
step with `allreduce_grads`.  RCCL refuses two ranks on one device, so the transport here is gloo with device
tensors; everything else (ray-range shards generated on the device, per-shard jitter rows, padding of uneven
shards, frame reassembly, gradient averaging) is the code the multi-GPU bench executes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import synth  # noqa: E402

W, H, NC, NF = 37, 19, 16, 24          # 703 rays: odd, so the two shards differ in size


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _models(dev):
    from mirender import fields
    cm = fields.field_from_state_dict(synth.state_dict("nerf", seed=0, sharp=True, bias_jitter=0.05), dev)
    fm = fields.field_from_state_dict(synth.state_dict("nerf", seed=1, sharp=True, bias_jitter=0.05), dev)
    return cm, fm


def _frame(dev, group_ok):
    from mirender import dist as mdist
    cm, fm = _models(dev)
    pose = synth.pose_degrees(4.0, 20.0, -30.0)
    tr = synth.t_rand(W * H, NC, seed=2).to(dev)
    out = mdist.render_image_dist(W, H, 1.3875 * W, pose, 2.0, 6.0, cm, fm, NC, NF, t_rand=tr)
    seeded = mdist.render_image_dist(W, H, 1.3875 * W, pose, 2.0, 6.0, cm, fm, NC, NF, seed=77)
    return [t.cpu().numpy() for t in out], [t.cpu().numpy() for t in seeded], (cm, fm)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirender import dist as mdist, render_core, train
    frame, seeded, (cm, fm) = _frame(dev, True)
    # data-parallel training step: each rank its own half of a 128-ray batch, gradients averaged
    rays = torch.from_numpy(__import__("oracle").render_ref.rays_from_camera(W, H, 1.3875 * W, synth.pose_degrees(4.0, 20.0, -30.0))[:128]).to(dev)
    tr = synth.t_rand(128, NC, seed=3).to(dev)
    tgt = torch.rand((128, 4), generator=torch.Generator().manual_seed(5)).to(dev)
    a, b = mdist.shard_range(128, rank, world)
    params = list(cm.parameters()) + list(fm.parameters())
    outs = render_core.render_rays(rays[a:b], 2.0, 6.0, cm, fm, NC, NF, t_rand=tr[a:b])
    loss, _ = train.nerf_loss(outs, tgt[a:b, :3], tgt[a:b, 3], use_alpha=True, use_fine_model=True)
    loss.backward()
    mdist.allreduce_grads(params)
    out[rank] = (frame, seeded, [p.grad.cpu().numpy() for p in params])
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_device_frame_and_training_step():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    dev = torch.device("cuda", 0)
    single, single_seeded, (cm, fm) = _frame(dev, False)           # no process group: the whole frame on one rank
    from mirender import render_core, train
    from oracle import render_ref as R
    rays = torch.from_numpy(R.rays_from_camera(W, H, 1.3875 * W, synth.pose_degrees(4.0, 20.0, -30.0))[:128]).to(dev)
    tr = synth.t_rand(128, NC, seed=3).to(dev)
    tgt = torch.rand((128, 4), generator=torch.Generator().manual_seed(5)).to(dev)
    params = list(cm.parameters()) + list(fm.parameters())
    outs = render_core.render_rays(rays, 2.0, 6.0, cm, fm, NC, NF, t_rand=tr)
    loss, _ = train.nerf_loss(outs, tgt[:, :3], tgt[:, 3], use_alpha=True, use_fine_model=True)
    loss.backward()
    for rank in range(world):
        frame, seeded, grads = out[rank]
        for got, ref in zip(frame, single):
            assert got.shape == ref.shape and np.array_equal(got, ref)          # injected jitter: bit-exact
        for got, ref in zip(seeded, single_seeded):
            assert np.array_equal(got, ref)                                     # Philox keyed by absolute ray index
        # mean of the two half-batch gradients == the full-batch gradient (mean losses over equal halves)
        for g, p in zip(grads, params):
            ref = p.grad.cpu().numpy()
            assert np.abs(g - ref).max() <= 2e-5 * max(1e-3, np.abs(ref).max())


@pytest.mark.timeout(600)
def test_the_drivers_two_gpu_command_end_to_end():
    """`python bench.py --gpus 2 ...` exactly as the driver's scaling run starts it (the script spawns its own two ranks through
    torch.distributed.run), with both ranks on this box's one device (MI_BENCH_REHEARSAL=2: gloo, the pi_GAN images shrunk so that
    two steps fit next to whatever this test process holds): ONE JSON line comes back, the frame was ray-sharded over two ranks, and
    its `train` object holds the nerf step and the C5 step with two ranks' step times and gradient all-reduce times."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MI_BENCH_REHEARSAL"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-frame64"],
                       env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["collective"]["n_ranks_seen"] == 2
    assert len(d["collective"]["per_rank_allgather_ms"]) == 2 and d["value"] > 0
    for wl in ("nerf_train", "c5"):
        t = d["train"][wl]
        assert t["n_ranks_seen"] == 2 and t["scaling"] == "weak" and t["ms_per_step"] > 0
        assert len(t["per_rank_ms_per_step"]) == 2 and len(t["per_rank_grad_allreduce_ms"]) == 2
        assert all(x is not None and x > 0 for x in t["per_rank_grad_allreduce_ms"])
        assert t["grad_allreduce_bytes"] in (4751392, 7643152)
        assert 0 < t["frac"] <= t["frac_reference_equivalent"] * (1 + 1e-12) <= 1.0 + 1e-9
    assert "c4" not in d["train"]                       # the batch-32 single-GPU configuration stays at N = 1
