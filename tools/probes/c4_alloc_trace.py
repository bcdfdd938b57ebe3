"""Diagnostic (not part of the product): does the caching allocator go back to the driver inside a C4 / C5 training step?
Per step: wall time, hipMalloc / hipFree calls torch made (num_device_alloc / num_device_free), allocation retries
(a failed hipMalloc followed by a flush of the cache), the driver's free figure and torch's reserved / allocated bytes;
with MI_DEBUG_PLAN=1 also what the memory planner of mirender/autograd.py decided per pass.
    python tools/probes/c4_alloc_trace.py [c4|c5] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
import torch  # noqa: E402

from mirender import dist as mdist, pigan  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(0)
res, b, nc, nf = (256, 4, 24, 48) if wl == "c5" else (128, 32, 12, 24)
gen = pigan.Generator(256, res, near=0.5, far=1.5, fov=12, coarse_samples=nc, fine_samples=nf).to(dev)
params = list(gen.parameters())
opt = torch.optim.Adam(params, lr=5e-5, betas=(0.0, 0.9))
z = torch.randn(b, 256, device=dev)
free, total = torch.cuda.mem_get_info(dev)
print(f"device memory: total {total / 2**30:.1f} GiB, free at start {free / 2**30:.1f} GiB", flush=True)
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")
last = {k: 0 for k in keys}
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if wl == "c5":
        with torch.no_grad():
            gen(z, seed=500 + i)
    img = gen(z, seed=100 + i)
    t1 = time.perf_counter()
    loss = torch.nn.functional.softplus(-img.mean(dim=(1, 2, 3))).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t2 = time.perf_counter()
    mdist.allreduce_grads(params)
    opt.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    st = torch.cuda.memory_stats(dev)
    now = {k: st.get(k, 0) for k in keys}
    free, _ = torch.cuda.mem_get_info(dev)
    print(f"step {i}: {1e3 * (t3 - t0):7.1f} ms (host: forward issued after {1e3 * (t1 - t0):6.1f}, backward after {1e3 * (t2 - t0):6.1f}) | "
          + " ".join(f"{k[4:]} +{now[k] - last[k]}" for k in keys)
          + f" | driver free {free / 2**30:6.1f} GiB, reserved {st['reserved_bytes.all.current'] / 2**30:6.1f}, "
            f"allocated {st['allocated_bytes.all.current'] / 2**30:5.1f}, peak allocated {st['allocated_bytes.all.peak'] / 2**30:6.1f}, "
            f"inactive split {st['inactive_split_bytes.all.current'] / 2**30:5.1f} GiB", flush=True)
    last = now
