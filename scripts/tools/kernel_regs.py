"""Registers, spills and scratch of every kernel in the objects of the library build (uc_nerf_amd/csrc/_obj/*.o), from the code objects' metadata.

  python scripts/tools/kernel_regs.py [object stems ...] [--all]     default: kernels that spill or use scratch, and every gather-fused instantiation
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "uc_nerf_amd", "csrc", "_obj")


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "a.fatbin"), os.path.join(d, "a.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        ids = subprocess.run([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + fat], capture_output=True, text=True, check=True).stdout.split()
        tgt = next(t for t in ids if "gfx950" in t)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + tgt, "--input=" + fat, "--output=" + co], check=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, k).group(1)      # noqa: E731
        out.append(dict(name=g("name"), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), spill=int(g("vgpr_spill_count")),
                        scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size"))))
    return out


if __name__ == "__main__":
    show_all = "--all" in sys.argv
    stems = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(f[:-2] for f in os.listdir(OBJ) if f.endswith(".o") and f != "source_hash.o")
    for stem in stems:
        for k in kernels(os.path.join(OBJ, stem + ".o")):
            name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            if show_all or k["spill"] or k["scratch"] or "true, true" in name.replace("(bool)1", "true"):
                print("%-18s vgpr %3d sgpr %3d spill %3d scratch %4d lds %6d  %s" % (stem, k["vgpr"], k["sgpr"], k["spill"], k["scratch"], k["lds"], name[:150]))
