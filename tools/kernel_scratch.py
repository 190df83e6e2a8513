#!/usr/bin/env python3
"""Scratch memory per kernel of a built library / object: VGPRs, spilled VGPRs / SGPRs, private segment bytes per lane and LDS,
read from the code objects' metadata (no recompilation).

    python3 tools/kernel_scratch.py [basevar_amd/lib/libbasevar_amd.so]

Why it exists: round 5 found 432 B of scratch per lane in bv_p1s_fused_kernel -- callee-saved registers stored around a call
clang had marked `tail`, and loop invariants LLVM had hoisted out of the persistent loop and the register allocator then
spilled -- i.e. 130 MB of HBM writes per launch and a memory trip per reload inside the solver's dependent chains.
tests/test_abi_cpu.py holds the product kernels to the figures this prints.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    """the gfx950 ELF images embedded in a host library / object (or the file itself when it is one)"""
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(b"\x7fELF", pos)
        if pos < 0:
            return
        if struct.unpack_from("<H", data, pos + 18)[0] == 224:  # EM_AMDGPU
            shoff = struct.unpack_from("<Q", data, pos + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, pos + 58)
            yield data[pos:pos + shoff + shentsize * shnum]
        pos += 4


def kernels(path):
    """[{name, vgpr, vgpr_spill, sgpr_spill, private, lds}] over every kernel of every embedded code object"""
    out = []
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as fh:
            fh.write(img)
            fh.flush()
            notes = subprocess.run([READELF, "--notes", fh.name], capture_output=True, text=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
            out.append({"name": re.search(r"\.name:\s+(\S+)", block).group(1), "vgpr": g("vgpr_count"), "vgpr_spill": g("vgpr_spill_count"),
                        "sgpr_spill": g("sgpr_spill_count"), "private": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")})
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return [re.sub(r"\(.*", "", l.replace("void ", "")) for l in r.stdout.split("\n")][:len(names)]
    except OSError:
        return names


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "basevar_amd", "lib", "libbasevar_amd.so")
    ks = kernels(lib)
    for k, n in zip(ks, demangle([k["name"] for k in ks])):
        print("%-64s vgpr %3d  spilled vgpr %3d sgpr %3d  private %4d B/lane  lds %6d" % (n[:64], k["vgpr"], k["vgpr_spill"], k["sgpr_spill"], k["private"], k["lds"]))
