#!/usr/bin/env python3
"""CPU probe (test infrastructure): in which training regime is the reference loop itself reproducible?

For one (student, optimiser, lr, batch, steps) the loop of oracle/fit_ref.py runs on the CPU in fp32, in fp64, and
(third leg) in fp32 from initial weights perturbed by 1e-6 relative - about what separates two fp32 implementations
of the forward pass - and the loss curves / held-out PSNRs are compared.  A regime is usable as a HARD 0.05 dB / 1 %
gate for another fp32 implementation only if BOTH differences are an order of magnitude below the gate
(tests/test_gpu_psnr.py): Adam at 5e-5 passed the precision leg (1e-4 dB) and failed the perturbation leg.

    python tools/probes/fit_regimes.py siren_nerf adam 1e-5 0 15
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import fit_ref  # noqa: E402


def main():
    student, optimizer, lr0, batch, steps = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    torch.set_num_threads(int(os.environ.get("FIT_THREADS", os.cpu_count())))
    scene = fit_ref.Scene(student=student)
    t0 = time.time()
    l32, p32, _ = fit_ref.fit_cpu(scene, steps, batch, lr0=lr0, optimizer=optimizer)
    t1 = time.time()
    l64, p64, _ = fit_ref.fit_cpu(scene, steps, batch, f64=True, lr0=lr0, optimizer=optimizer)
    l32, l64 = np.array(l32), np.array(l64)
    rel = np.abs(l32 - l64) / l64
    print(f"{student} {optimizer} lr {lr0:g} batch {batch or 'all'} steps {steps}: loss {l64[0]:.5f} -> {l64[-1]:.5f} "
          f"(x{l64[-1] / l64[0]:.3f}); fp32 vs fp64: max rel loss {rel.max():.2e} (at step {int(rel.argmax())}), "
          f"psnr {p32:.4f} vs {p64:.4f} dB (|d| {abs(p32 - p64):.4f}); {t1 - t0:.0f}s fp32 / {time.time() - t1:.0f}s fp64",
          flush=True)
    print("  rel per step:", " ".join(f"{r:.1e}" for r in rel), flush=True)
    rng = np.random.Generator(np.random.PCG64(9))
    for sd in scene.student_init:
        for k in sd:
            sd[k] = sd[k] * torch.from_numpy((1 + 1e-6 * rng.standard_normal(sd[k].shape)).astype(np.float32))
    lp, pp, _ = fit_ref.fit_cpu(scene, steps, batch, lr0=lr0, optimizer=optimizer)
    relp = np.abs(np.array(lp) - l32) / l32
    print(f"  initial weights perturbed by 1e-6 relative (fp32): max rel loss {relp.max():.2e}, psnr {pp:.4f} dB (|d| {abs(pp - p32):.4f})",
          flush=True)


if __name__ == "__main__":
    main()
