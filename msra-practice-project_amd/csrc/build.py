#!/usr/bin/env python3
"""Build libmirender.so for gfx950 with hipcc (in-tree, no JIT cache).

    python msra-practice-project_amd/csrc/build.py [--force]

Outputs msra-practice-project_amd/mirender/libmirender.so next to the Python binding so it
travels to the GPU box with the source snapshot.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "mirender", "libmirender.so")
OBJ = os.path.join(HERE, "_obj")
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
SOURCES = {
    "field_mlp.hip": [],
    "field_mlp_bwd.hip": [],
    # un-fused mul/add like the torch / NumPy ops these kernels restate
    "render_stages.hip": ["-ffp-contract=off"],
    "eval_stages.hip": ["-ffp-contract=off"],
    "adam_step.hip": [],
    "api.hip": [],
}
HEADERS = ["field_layout.h", "mi_common.h", "mi_math.h", "field_mlp_device.h", os.path.join("..", "..", "include", "mi_render.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_profile(verbose=True):
    """Diagnostic build with in-kernel cycle stamps (-DMI_PROFILE_STAMPS): gpurun_tools/libmirender_prof.so.
    Never loaded by the product or the tests; used by tools/stamp_profile.py only."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(os.path.dirname(PKG), "gpurun_tools")
    os.makedirs(out_dir, exist_ok=True)
    extra = os.environ.get("MI_EXTRA_DEFS", "").split()          # extra -D switches for one-off diagnostic builds
    out = os.path.join(out_dir, os.environ.get("MI_PROF_NAME", "libmirender_prof.so"))
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    cmd = [hipcc, *COMMON, "-DMI_PROFILE_STAMPS", *extra, "-shared", *srcs, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, timeout=2400)
    return out


def build(force=False, verbose=True, always=()):
    """Compile what is stale (everything with force; the translation units named in `always` regardless) and link.
    Prints, per translation unit, whether it was COMPILED or REUSED - the record of what a given run of the build
    check really exercised - and returns the library's path."""
    import time
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs, jobs = [], []
    t_start = time.time()
    for src, extra in SOURCES.items():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        why = "forced" if force else "always rebuilt by the build check" if src in always else \
            "no object yet" if not os.path.exists(o) else "source or header newer than the object" if _stale(o, [s] + hdrs) else None
        if why:
            jobs.append([hipcc, *COMMON, *extra, "-c", s, "-o", o])
            print(f"[build] {src}: COMPILED for {ARCH} ({why})", flush=True)
        else:
            print(f"[build] {src}: REUSED {os.path.relpath(o, PKG)} (newer than its source and every header)", flush=True)
    # the translation units are independent (the two MLP files take minutes each): compile them side by side
    from concurrent.futures import ThreadPoolExecutor

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, timeout=1800)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(run, jobs))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        print(f"[build] {os.path.relpath(OUT, PKG)}: LINKED", flush=True)
    else:
        print(f"[build] {os.path.relpath(OUT, PKG)}: REUSED (newer than every object)", flush=True)
    print(f"[build] {len(jobs)} of {len(SOURCES)} translation units compiled in {time.time() - t_start:.0f} s", flush=True)
    return OUT


def build_host_example(verbose=True):
    """tests/cabi/cabi_host: a C++ program that uses the library through include/mi_render.h alone (no Python, no torch) -
    the link line a C / C++ host of the ABI uses.  Test infrastructure (tests/test_gpu_cabi_host.py runs it)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    root = os.path.dirname(PKG)
    src = os.path.join(root, "tests", "cabi", "cabi_host.cpp")
    out = os.path.join(root, "tests", "cabi", "cabi_host")
    if _stale(out, [src, OUT, os.path.join(root, "include", "mi_render.h")]):
        cmd = [hipcc, "-O2", "-std=c++17", "-Wall", "-I", os.path.join(root, "include"), src, "-L", os.path.dirname(OUT), "-lmirender",
               "-Wl,-rpath,$ORIGIN/../../msra-practice-project_amd/mirender", "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, timeout=600)
        print(f"[build] {os.path.relpath(out, root)}: LINKED against libmirender.so", flush=True)
    else:
        print(f"[build] {os.path.relpath(out, root)}: REUSED", flush=True)
    return out


if __name__ == "__main__":
    if "--profile" in sys.argv:
        print(build_profile())
    else:
        print(build(force="--force" in sys.argv))
        print(build_host_example())
