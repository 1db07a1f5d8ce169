"""Builds libucnerf_hip.so (gfx950 only) in-tree with hipcc.  `python -m uc_nerf_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the working tree.
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libucnerf_hip.so")
SOURCES = ["rays.hip", "gather.hip", "gather_cl.hip", "mlp.hip", "mlp_bf16.hip", "mlp_bwd.hip", "mlp_bwd_chain.hip", "mlp_wgrad.hip", "composite.hip", "sample_pdf.hip", "render.hip", "mvs.hip"]
# (source, object name, extra flags): translation units built more than once with different switches
VARIANTS = [("mlp_bf16.hip", "mlp_bf16_plain.o", ["-DUCNERF_BF16_BUILD_TERMS=1"]), ("mlp_bf16.hip", "mlp_bf16_tail.o", ["-DUCNERF_BF16_BUILD_TAIL=1"]),
            # the same three builds with fp16 terms (ucnerf_mlp_config.operand == 1, ABI v6): entry points under the suffix _h16
            ("mlp_bf16.hip", "mlp_h16.o", ["-DUCNERF_OPERAND_FP16=1"]),
            ("mlp_bf16.hip", "mlp_h16_plain.o", ["-DUCNERF_OPERAND_FP16=1", "-DUCNERF_BF16_BUILD_TERMS=1"]),
            ("mlp_bf16.hip", "mlp_h16_tail.o", ["-DUCNERF_OPERAND_FP16=1", "-DUCNERF_BF16_BUILD_TAIL=1"])]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mlp_layout.h"), os.path.join(CSRC, "sincos_cw.h"), os.path.join(CSRC, "gather_cl_device.h"), os.path.join(CSRC, "mlp_bwd_parts.h"), os.path.join(CSRC, "p24.h"), os.path.join(CSRC, "composite_device.h"), os.path.join(CSRC, "sample_pdf_device.h"), os.path.join(CSRC, "raygen_device.h"),
           os.path.join(HERE, "..", "include", "ucnerf_hip.h")]
# -ffp-contract=off: the sample_pdf / sampling kernels reproduce torch-CPU roundings (separate mul and add)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


STAMP = os.path.join(OBJ, "built_here.json")


def source_hash():
    """sha256 over everything the library is made of: the translation units, the headers, the flags and the variant list.  The linker
    step compiles it into the library (`ucnerf_source_hash()`), so a binary can be matched against the tree it sits in."""
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(x) for x in HEADERS]):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS, VARIANTS)).encode())
    return h.hexdigest()[:32]


def _machine():
    """Identity of the machine a build ran on (a snapshot of the tree carried to another box keeps the stamp but not this)."""
    parts = []
    for path in ("/proc/sys/kernel/random/boot_id", "/etc/hostname"):
        try:
            with open(path) as f:
                parts.append(f.read().strip())
        except OSError:
            parts.append("?")
    return "|".join(parts)


def built_here():
    """True when the library on disk was linked on THIS machine from the sources as they are now."""
    try:
        with open(STAMP) as f:
            st = json.load(f)
    except (OSError, ValueError):
        return False
    return os.path.exists(LIB) and st.get("machine") == _machine() and st.get("hash") == source_hash()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    for s, oname, extra in VARIANTS:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, oname)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=6) as ex:
        for err in ex.map(run, jobs):
            if verbose and err.strip():
                print(err)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs] + [os.path.join(OBJ, o) for _, o, _ in VARIANTS]
    digest = source_hash()
    stamp_c, stamp_o = os.path.join(OBJ, "source_hash.c"), os.path.join(OBJ, "source_hash.o")
    text = '/* generated by uc_nerf_amd/build.py */\nconst char* ucnerf_source_hash(void) { return "%s"; }\n' % digest
    if not os.path.exists(stamp_c) or open(stamp_c).read() != text:
        with open(stamp_c, "w") as f:
            f.write(text)
    if force or _stale(stamp_o, [stamp_c]):
        run([hipcc, "-O1", "-fPIC", "-x", "c", "-c", stamp_c, "-o", stamp_o])
    objs.append(stamp_o)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        with open(STAMP, "w") as f:
            json.dump({"machine": _machine(), "hash": digest}, f)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
