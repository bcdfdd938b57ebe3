#!/usr/bin/env python3
"""The loop of pi_GAN/train.py (lines 48-56, 91-172) on the MI355X path, end to end and without a dataset.

What the reference's script does per iteration - D step on real + generated images with the R1 penalty, G step through
the renderer, both learning rates decayed, progressive resolution stages with `set_resolution`, checkpoints of the
script's own dict - written against the product: `mirender.pigan.Generator` (the drop-in for pi_GAN/modules.py:165-197:
mapping network on PyTorch-ROCm, the whole batch rendered by the fused FiLM kernels, gradients back to the mapping
network), `requires_grad(...)` toggling like train.py:100-101,123-124 (a generator whose parameters do not require grad
takes the inference kernels), `mirender.checkpoint.save_pigan`, and - replacing the script's torch.nn.DataParallel
(train.py:50,52) - one process per GPU with ONE flat gradient all-reduce per network and step
(`mirender.dist.allreduce_grads`):

    python examples/train_pigan_synthetic.py [--steps 60] [--stages 16,32] [--batch 8]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_pigan_synthetic.py ...

No dataset ships with the reference (pi_GAN/dataloader.py reads CelebA-style folders) and none can be fetched, so the
"photographs" are renders of a fixed TEACHER generator (another seed) at random latents and poses.  The discriminator is
a small stock-PyTorch CNN: the reference's CoordConv discriminator (modules.py:205-330) is outside the render path
 and any image critic exercises the same generator calls.  Only the product is imported: nothing
from oracle/."""
import argparse
import os
import sys
import tempfile
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "msra-practice-project_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf-only hosts: RCCL's peer-to-peer setup needs it

from mirender import checkpoint, dist as mdist, pigan  # noqa: E402


class SmallDiscriminator(torch.nn.Module):
    """Any image critic will do here (see the module docstring); resolution-agnostic through adaptive pooling, so the
    progressive stages of train.py:83-86,146-152 need no `set_resolution` of its own."""

    def __init__(self, width=32):
        super().__init__()
        self.net = torch.nn.Sequential(
            torch.nn.Conv2d(3, width, 3, padding=1), torch.nn.LeakyReLU(0.2),
            torch.nn.Conv2d(width, 2 * width, 3, stride=2, padding=1), torch.nn.LeakyReLU(0.2),
            torch.nn.Conv2d(2 * width, 4 * width, 3, stride=2, padding=1), torch.nn.LeakyReLU(0.2),
            torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(4 * width, 1))

    def forward(self, x):
        return self.net(x)


def requires_grad(module, flag):                       # pi_GAN/utils.py's helper, as train.py:100-101 uses it
    for p in module.parameters():
        p.requires_grad_(flag)


def loss_f(u):                                         # pi_GAN/utils.py:28-29
    return -F.softplus(-u)


def loss_r1(y, x):                                     # the R1 penalty of train.py:117 (pi_GAN/utils.py:32-37)
    g, = torch.autograd.grad(y, [x], torch.ones_like(y), create_graph=True)
    return torch.mean(g.reshape(x.shape[0], -1).norm(dim=-1) ** 2)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60, help="iterations per stage")
    ap.add_argument("--stages", default="16,32", help="progressive resolutions (train.py's `resolution` list)")
    ap.add_argument("--batch", type=int, default=8, help="images per process and step")
    ap.add_argument("--z-dim", type=int, default=64)
    ap.add_argument("--samples", default="12,24", help="coarse,fine samples per ray (train.py:25-26)")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    assert torch.cuda.is_available(), "needs a ROCm device: there is no CPU path"
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)      # = RCCL on ROCm
    stages = [int(s) for s in args.stages.split(",")]
    nc, nf = (int(s) for s in args.samples.split(","))
    g_lr, d_lr, g_lr_end, d_lr_end, lr_decay = 5e-5, 4e-4, 1e-5, 1e-4, 500              # train.py:34-38

    # identical replicas on every rank (what DataParallel's broadcast gives the reference), own data per rank
    torch.manual_seed(0)
    generator = pigan.Generator(args.z_dim, stages[0], 0.5, 1.5, 12, nc, nf, 0.45, 0.15, True).to(dev)      # train.py:49
    discriminator = SmallDiscriminator().to(dev)
    teacher = pigan.Generator(args.z_dim, stages[0], 0.5, 1.5, 12, nc, nf, 0.45, 0.15, True).to(dev)        # the "dataset"
    requires_grad(teacher, False)
    g_optimizer = torch.optim.Adam(generator.parameters(), lr=g_lr, betas=(0.0, 0.9))                        # train.py:53-54 (torch >= 2.x wants both betas as floats)
    d_optimizer = torch.optim.Adam(discriminator.parameters(), lr=d_lr, betas=(0.0, 0.9))
    torch.manual_seed(1000 + rank)
    loss_log = {"g_loss": [], "d_loss": []}
    global_step, t0 = 0, time.time()

    for stage, res in enumerate(stages):
        generator.set_resolution(res)                                                                        # train.py:85,150
        teacher.set_resolution(res)
        for _ in range(args.steps):
            global_step += 1
            with torch.no_grad():                       # dataset.get(): a batch of "photographs" [b,3,H,W] in [0,1]
                real_image = teacher(torch.randn(args.batch, args.z_dim, device=dev), seed=global_step)
            # ---- train D (train.py:99-121)
            requires_grad(generator, False)
            requires_grad(discriminator, True)
            real_image = real_image.clone().requires_grad_(True)
            real_label = discriminator(real_image)
            z = torch.randn(args.batch, args.z_dim, device=dev)
            gen_image = generator(z)                    # no parameter requires grad: the inference kernels, no saved rows
            assert not gen_image.requires_grad
            gen_label = discriminator(gen_image)
            d_optimizer.zero_grad()
            d_loss = -torch.mean(loss_f(gen_label)) - torch.mean(loss_f(-real_label)) + 1.0 * loss_r1(real_label, real_image)    # :117
            d_loss.backward()
            mdist.allreduce_grads(discriminator.parameters())
            d_optimizer.step()
            # ---- train G (train.py:123-136)
            requires_grad(generator, True)
            requires_grad(discriminator, False)
            z = torch.randn(args.batch, args.z_dim, device=dev)
            gen_image = generator(z)                    # training kernels: saving forward, fused backward
            gen_label = discriminator(gen_image)
            g_optimizer.zero_grad()
            g_loss = torch.mean(loss_f(gen_label))                                                                        # :133
            g_loss.backward()
            mdist.allreduce_grads(generator.parameters())
            g_optimizer.step()
            loss_log["d_loss"].append(float(d_loss.detach()))
            loss_log["g_loss"].append(float(g_loss.detach()))
            # ---- learning rates (train.py:138-147)
            decay = 0.1 ** (global_step / (lr_decay * 1000))
            for grp in g_optimizer.param_groups:
                grp["lr"] = g_lr_end + (g_lr - g_lr_end) * decay
            for grp in d_optimizer.param_groups:
                grp["lr"] = d_lr_end + (d_lr - d_lr_end) * decay
            if not args.quiet and rank == 0 and global_step % 20 == 0:
                print(f"[stage {stage} {res}x{res}] iter {global_step}: d_loss {loss_log['d_loss'][-1]:.4f} g_loss {loss_log['g_loss'][-1]:.4f}", flush=True)
    torch.cuda.synchronize()
    elapsed = time.time() - t0

    # the script's checkpoint (train.py:160-172), written and read back; the restored generator paints the same image
    result = {"steps": global_step, "seconds": elapsed, "d_loss": loss_log["d_loss"], "g_loss": loss_log["g_loss"],
              "image_shape": tuple(gen_image.shape), "world": world}
    if rank == 0:
        with tempfile.TemporaryDirectory() as tmp:
            path = checkpoint.checkpoint_path(tmp, global_step)
            checkpoint.save_pigan(path, global_step, loss_log, generator, discriminator, g_optimizer, d_optimizer)
            ck = checkpoint.load(checkpoint.latest(tmp))
            restored = checkpoint.pigan_generator(ck, stages[-1], dev, near=0.5, far=1.5, fov=12, coarse_samples=nc, fine_samples=nf,
                                                  horizontal_std=0.45, vertical_std=0.15)
            z = torch.randn(2, args.z_dim, device=dev)
            with torch.no_grad():
                a = generator(z, thetas=[0.1, -0.2], phis=[0.0, 0.1], seed=7)
                b = restored(z, thetas=[0.1, -0.2], phis=[0.0, 0.1], seed=7)
            result["checkpoint_roundtrip_max_abs"] = float((a - b).abs().max())
            result["global_step_restored"] = int(ck["global_step"])
        if not args.quiet:
            print(f"{global_step} iterations in {elapsed:.1f} s ({elapsed / global_step * 1e3:.1f} ms each, batch {args.batch} per process, "
                  f"{world} process(es)); checkpoint round trip |diff| {result['checkpoint_roundtrip_max_abs']:.1e}", flush=True)
    if dist.is_initialized() and world > 1:
        dist.barrier()
    return result


if __name__ == "__main__":
    main()
