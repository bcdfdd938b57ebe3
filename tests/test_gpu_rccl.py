"""GPU: the product's RCCL calls executed for real, on one GPU (VERDICT r02 "next" #3).

`render_image_dist`'s all_gather_into_tensor and `allreduce_grads`' flat all_reduce return early when a group has one
rank; with mirender.dist.FORCE_COLLECTIVE they are issued anyway.  tests/rccl_one_rank.py - started here as a FRESH child
process (it initialises its own `nccl` group; this process, which has already used the GPU, is never re-exec'ed) -
renders a sharded frame, a data-parallel nerf step and a pi_GAN generator step through a one-rank RCCL group and
compares each bit for bit with the ungrouped result.  Its output is kept as gpurun_out/r04_rccl_1rank.log (committed
copy: profiles/r04_rccl_1rank.log; round 3: profiles/r03_rccl_1rank.log)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_one_rank_rccl_group_runs_every_collective_of_the_product():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=800)
    log = f"$ NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL python tests/rccl_one_rank.py   (exit {r.returncode})\n" \
          f"--- stdout ---\n{r.stdout}\n--- stderr ---\n{r.stderr[-20000:]}\n"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_rccl_1rank.log"), "w") as f:
        f.write(log)
    assert r.returncode == 0, log[-4000:]
    assert "backend nccl" in r.stdout and "rccl one-rank: OK" in r.stdout
    assert r.stdout.count("PASS bit-equal") == 3


@pytest.mark.timeout(600)
def test_bench_force_collective_runs_the_headline_frame_through_a_one_rank_rccl_group():
    """`python bench.py --force-collective` (N = 1): the timed 800x800 frame is assembled by RCCL's all_gather_into_tensor
    in a one-rank nccl group, and the line carries the `collective` object the N > 1 lines carry - the code path of the
    driver's 8-GPU scaling run, exercised where only one GPU exists."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-collective", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-frame64", "--no-train"], capture_output=True, text=True, timeout=500,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["collective"]
    assert c["backend"] == "nccl" and c["n_ranks_seen"] == 1 and c["frames"] == 1
    assert c["allgather_bytes_per_rank"] == 800 * 800 * 5 * 4 and c["per_rank_allgather_ms"][0] > 0
    assert line["n_gpus"] == 1 and line["value"] > 1e5
