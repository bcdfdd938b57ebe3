#!/usr/bin/env python3
"""Build container: where does a kernel wait on an LDS read it has only just issued?

    python tools/isa_lds_waits.py msra-practice-project_amd/csrc/_obj/field_mlp_bwd.o [kernel-name-filter]

For every `s_waitcnt` that waits for LDS (lgkmcnt(N)) the script works out which ds_read it really waits for - the
(N+1)-th youngest outstanding one - and estimates the issue cycles between that read and the wait (MFMA 64, transcendental
10, packed 8, other VALU 5, LDS 12, VMEM 16, scalar 1: the figures of tools/probes/mfma_valu_overlap.hip).  With one wave
per SIMD nothing else fills the time: a wait with less than ~64 cycles of work behind its read idles the SIMD for the rest
of the LDS latency.  The dW GEMM of rounds 1-2 did that once per step of 16 MFMAs (read, s_waitcnt lgkmcnt(0), MFMAs): the
round-3 fix came from this listing.  Straight-line estimate: loops are scanned once, branches ignored.
Prints, per kernel, a histogram {cycles behind the read, in buckets of 32: count}; the first two buckets are the suspects."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        asm = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True).split("\n")
    starts = [(i, l) for i, l in enumerate(asm) if re.match(r"^[0-9a-f]+ <", l)] + [(len(asm), "")]
    for (i, l), (j, _) in zip(starts, starts[1:]):
        name = l.split("<")[1].rstrip(">:")
        if flt not in name:
            continue
        mfma = 0                       # estimated issue cycles so far
        pending = []                   # MFMA count at the issue of every outstanding LDS / scalar-memory op (lgkm counter)
        hist = collections.Counter()
        for x in asm[i:j]:
            m = re.match(r"\s+(\S+)\s*(.*?)\s*//", x)
            if not m:
                continue
            op, args = m.group(1), m.group(2)
            cost = 64 if op.startswith("v_mfma") else 10 if re.match(r"v_(sin|cos|sqrt|rcp|rsq|exp|log)", op) else \
                8 if op.startswith("v_pk_") else 5 if op.startswith("v_") else 12 if op.startswith("ds_") else \
                16 if re.match(r"(buffer|global|scratch|flat)_", op) else 1
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                pending.append((mfma + cost, op.startswith("ds_read")))
            elif op == "s_waitcnt":
                w = re.search(r"lgkmcnt\((\d+)\)", args)
                if w:
                    keep = int(w.group(1))
                    done, pending = pending[:len(pending) - keep], pending[len(pending) - keep:] if keep else []
                    if done and done[-1][1]:
                        hist[min((mfma - done[-1][0]) // 32, 8)] += 1
            mfma += cost
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:64]
        short = sum(v for k, v in hist.items() if k < 2)
        print(f"{dem:66s} waits on LDS reads {sum(hist.values()):4d}, with < 64 cycles behind the read: {short:4d}   "
              + " ".join(f"{32 * k}{'+' if k == 8 else ''}:{hist[k]}" for k in sorted(hist)))


if __name__ == "__main__":
    main()
