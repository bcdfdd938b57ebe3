#!/usr/bin/env python3
"""Build container: per-kernel instruction statistics of a compiled translation unit (object with a .hip_fatbin section).

    python tools/isa_stats.py msra-practice-project_amd/csrc/_obj/field_mlp_bwd.o [kernel-name-filter]

Prints, per kernel: instructions, MFMAs, v_accvgpr_read / write (AGPR<->VGPR moves: more reads than accumulator
registers means the register allocator is shuffling values through AGPRs), scratch loads / stores (spills: each reload
is followed by an s_waitcnt vmcnt(0) that drains every outstanding row load / store of the wave), s_nop, and the
register counts from the code object's metadata.  The round-3 finding that film_bwd_kernel spilled 25 VGPRs came from
here."""
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
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
        if os.environ.get("ISA_KEEP"):
            open(os.environ["ISA_KEEP"], "w").write("\n".join(asm))
    meta = {}
    for blk in notes.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name:
            meta[name.group(1)] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) for k in
                                   ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size") if re.search(rf"\.{k}:\s+(\d+)", blk)}
            meta[name.group(1)]["agpr_count"] = int(blk.split()[0])
    starts = [(i, l) for i, l in enumerate(asm) if re.match(r"^[0-9a-f]+ <", l)] + [(len(asm), "")]
    for (i, l), (j, _) in zip(starts, starts[1:]):
        name = l.split("<")[1].rstrip(">:")
        if flt not in name:
            continue
        c = collections.Counter()
        for x in asm[i:j]:
            m = re.match(r"\s+(\S+)", x)
            if m:
                c[m.group(1)] += 1
        sl = sum(v for k, v in c.items() if k.startswith("scratch_load"))
        ss = sum(v for k, v in c.items() if k.startswith("scratch_store"))
        mm = meta.get(name, {})
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:70]
        print(f"{dem:72s} instr {sum(c.values()):6d} mfma {c['v_mfma_f32_32x32x2_f32']:5d} accread {c['v_accvgpr_read_b32']:4d} "
              f"accwrite {c['v_accvgpr_write_b32']:4d} scratch ld/st {sl:3d}/{ss:3d} s_nop {c['s_nop']:5d} "
              f"vgpr {mm.get('vgpr_count', '?')} agpr {mm.get('agpr_count', '?')} spilled {mm.get('vgpr_spill_count', '?')}")


if __name__ == "__main__":
    main()
