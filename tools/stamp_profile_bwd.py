#!/usr/bin/env python3
"""Diagnostic: where a 128-point tile of the NeRF backward CHAIN spends its cycles (stamped build, never the product
library: python msra-practice-project_amd/csrc/build.py --profile -> gpurun_tools/libmirender_prof.so)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "msra-practice-project_amd")]
from mirender import _lib
_lib.LIB_PATH = os.path.join(ROOT, "gpurun_tools", os.environ.get("MI_PROF_NAME", "libmirender_prof.so"))
print("library:", _lib.LIB_PATH)
_lib.SIGNATURES["mi_debug_set_stamps"] = (None, [ctypes.c_void_p])
from mirender import autograd as A, fields, ops
lib = _lib.load()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
FILM = "film" in sys.argv[1:]        # python tools/stamp_profile_bwd.py film: the FiLM chain (film_bwd_kernel) instead of NeRF's
m = (fields.FilmSirenNeRF if FILM else fields.NeRF)().to(dev)
pf = fields.as_packed_field(m)
n, s = 256 * 24 * 128 // 64, 64                      # 24 tiles per CU
rays = torch.randn(n, 2, 3, device=dev); rays[:, 0] = torch.tensor([0., 0., 1. if FILM else 4.], device=dev); rays[:, 1, 2] = -1
z = torch.sort(torch.rand(n, s, device=dev) * (1 if FILM else 4) + (0.5 if FILM else 2), -1).values
film = (torch.rand((4, 9, 512), device=dev) + 0.5) if FILM else None
raw, saved = A._forward_pass(pf, rays, z, film, 1 << 40)
g_raw = torch.randn_like(raw)
tiles = n * s // 128
stamps = torch.zeros((tiles, 128), dtype=torch.int64, device=dev)
for rep in range(2):
    lib.mi_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    A._field_backward(pf, rays, z, raw, g_raw, film, dict(saved))
    torch.cuda.synchronize()
lib.mi_debug_set_stamps(None)
st = stamps.cpu().numpy().astype(np.int64)
body = st[512:]
if FILM:
    names = {1: "prologue: hidden_layer_rgb epilogue (C_8 decode)", 2: "layer 7 (sigma start)", 3: "layer 6", 4: "layer 5", 5: "layer 4",
             6: "layer 3", 7: "layer 2", 8: "layer 1", 9: "layer 0 (stores its own rows)"}
    total = np.median(body[:, 9] - body[:, 0])
    print(f"FiLM chain tile (after the head gradients): {total:.0f} cycles (median); ideal MFMA {8 * 65536}")
    for i in range(1, 10):
        d = np.median(body[:, i] - body[:, i - 1])
        extra = f"  ideal 65536  overhead {d - 65536:.0f} ({100 * (d - 65536) / 65536:.1f} %)" if i >= 2 else ""
        print(f"{i:2d} {names[i]:50s} {d:9.0f}  {100 * d / total:5.1f} %{extra}")
    rows = np.median(body[:, 33:65] - body[:, 32:64], axis=0)
    print("layer 4 rows (cycles, ideal 2048 each; K block = 4 rows, stage = 2 K blocks):")
    for kb in range(8):
        print(f"  K block {kb}: " + " ".join(f"{rows[4 * kb + r]:6.0f}" for r in range(4)))
    print(f"  sum {rows.sum():.0f} vs 65536")
    sys.exit(0)
names = {1: "prologue (heads, dir-layer epilogue)", 2: "dir1^T (4 K blocks, linear)", 3: "dir0^T + sigma (8)", 4: "L7^T (8)",
         5: "L6^T (8)", 6: "L5^T (8)", 7: "L4^T..L2^T (3 x 8)", 8: "L1^T (8)"}
ideal = {2: 4 * 8192, 3: 8 * 8192, 4: 8 * 8192, 5: 8 * 8192, 6: 8 * 8192, 7: 24 * 8192, 8: 8 * 8192}
total = np.median(body[:, 8] - body[:, 0])
print(f"tile total (median) {total:.0f} cycles; ideal MFMA {sum(ideal.values())}")
for i in range(1, 9):
    d = np.median(body[:, i] - body[:, i - 1])
    extra = f"  ideal {ideal[i]}  overhead {d - ideal[i]:.0f} ({100 * (d - ideal[i]) / ideal[i]:.1f} %)" if i in ideal else ""
    print(f"{i:2d} {names[i]:38s} {d:9.0f}  {100 * d / total:5.1f} %{extra}")
rows = np.median(body[:, 33:65] - body[:, 32:64], axis=0)
print("L7^T rows (cycles, ideal 2048 each; K block = 4 rows, stage = 2 K blocks):")
for kb in range(8):
    print(f"  K block {kb}: " + " ".join(f"{rows[4 * kb + r]:6.0f}" for r in range(4)))
print(f"  sum {rows.sum():.0f} vs 65536")
